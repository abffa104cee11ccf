#!/usr/bin/env python3
"""Same command line as the reference's dolfin/process_bench1.py (run from the repo root)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pfhubbenchmarks_amd.postprocess import main  # noqa: E402

if __name__ == "__main__":
    main()
