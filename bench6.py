#!/usr/bin/env python3
"""python bench6.py -- PFHub benchmark 6 on MI355X; writes results/bench6_out.csv (see pfhubbenchmarks_amd/drivers.py)."""
from pfhubbenchmarks_amd.drivers import main_bench6

if __name__ == "__main__":
    main_bench6()
