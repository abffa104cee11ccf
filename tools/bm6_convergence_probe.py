import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from pfhubbenchmarks_amd import verification as V
ts = (0.1, 0.3)
t0 = time.time()
fem, d = V.fem_be_limit(ts, dt=0.01, log=print, model="bm6")
print("fem limit", fem, d["h2_dt0"], d["h1_dt0"], "%.1f s" % (time.time() - t0)); t0 = time.time()
fd, d2 = V.fd_limit_bm6(ts, log=print)
print("fd limit", fd, d2, "%.1f s" % (time.time() - t0))
print("rel", np.abs(fd - fem) / np.abs(fem))
