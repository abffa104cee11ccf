"""Row-by-row relative distance between a driver's CSV and the reference's committed one (tests/golden/).
Usage: python tools/compare_csv.py results/bench2_out.csv tests/golden/bench2_out.csv"""
import sys

import numpy as np

a = np.loadtxt(sys.argv[1], delimiter=",", skiprows=1, ndmin=2)
b = np.loadtxt(sys.argv[2], delimiter=",", skiprows=1, ndmin=2)
n = min(len(a), len(b))
assert np.abs(a[:n, 0] - b[:n, 0]).max() < 1e-9, "time grids differ"
rf = np.abs(a[:n, 1] - b[:n, 1]) / np.abs(b[:n, 1])
rc = np.abs(a[:n, 2] - b[:n, 2]) / np.maximum(np.abs(b[:n, 2]), 1e-300)
ac = np.abs(a[:n, 2] - b[:n, 2])
print("%s vs %s: %d rows (of %d / %d), t = %g .. %g" % (sys.argv[1], sys.argv[2], n, len(a), len(b), a[0, 0], a[n - 1, 0]))
print("  column 2 (total_free_energy): max rel %.3e at row %d, median %.3e" % (rf.max(), int(rf.argmax()), np.median(rf)))
print("  column 3: max rel %.3e at row %d, max abs %.3e" % (rc.max(), int(rc.argmax()), ac.max()))
