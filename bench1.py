#!/usr/bin/env python3
"""python bench1.py -- PFHub benchmark 1 on MI355X; writes results/bench1_out.csv (see pfhubbenchmarks_amd/drivers.py)."""
from pfhubbenchmarks_amd.drivers import main_bench1

if __name__ == "__main__":
    main_bench1()
