#!/usr/bin/env python3
"""Shim so the reference's command line `python dolfin/bench2.py` (README.md:18-29, run from the repo root) works
verbatim against this repository."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pfhubbenchmarks_amd.drivers import main_bench2  # noqa: E402

if __name__ == "__main__":
    main_bench2()
