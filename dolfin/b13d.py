#!/usr/bin/env python3
"""Shim so the reference's command line `python dolfin/b13d.py` (3-D BM1) works against this repository."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pfhubbenchmarks_amd.drivers import main_b13d  # noqa: E402

if __name__ == "__main__":
    main_b13d()
