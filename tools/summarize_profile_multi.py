"""summary_<workload>.json for workloads whose step is SEVERAL kernels (spectral: 4 passes; BM6: Poisson passes + stencil;
BM2 / BM3 explicit FD) from the three rocprofv3 passes of tools/profile_workload.sh: every kernel that is launched once
per step (call count within 10 % of the largest), its average duration (kernel trace), FETCH_SIZE x 2 (gfx950 correction for
16-byte streaming reads, MI355X_MICROARCH.md) and WRITE_SIZE per launch; per step: the sums.  `traffic_bytes_per_launch` is
the per-STEP HBM traffic (bench.py's roofline.traffic reads it: a "launch" of these workloads is one step).
Usage: python tools/summarize_profile_multi.py <outdir> <workload> <tag>"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out, workload, tag = sys.argv[1], sys.argv[2], sys.argv[3]
CELLS = 512 ** 3
BYTES = {"bm1_spectral_512c": 72.0, "bm6_spectral_512c": 72.0, "bm6_fd_512c": 72.0, "bm6_fd_512c_elim": 16.0,
         "bm2_fd_512c": 80.0, "bm3_fd_512c": 32.0}[workload]


def one(pattern):
    f = sorted(glob.glob(os.path.join(out, pattern), recursive=True))
    if not f:
        sys.exit("missing " + pattern)
    return f[-1]


def short(name):
    name = name.replace("void pfhip::(anonymous namespace)::", "").replace("pfhip::(anonymous namespace)::", "")
    return name[:name.index("(")] if "(" in name else name


stats = list(csv.DictReader(open(one("stats/**/*kernel_stats.csv"))))
stats = [r for r in stats if "pfhip" in r["Name"]]
# kernels of the time loop: everything that takes a visible share of the GPU time; `steps` = launches of the least often
# launched one among them (a once-per-step kernel: the stencil, the z pass).  Since round 4 the plane-local passes of the
# spectral / Poisson steps run once per CHUNK of planes (17 launches per step at 512^3) on two streams: their per-launch
# averages are per chunk and overlap in time, so the sum of the kernel times exceeds the wall time of a step.
tot_max = max(float(r["TotalDurationNs"]) for r in stats)
step_kernels = [r for r in stats if float(r["TotalDurationNs"]) >= 0.15 * tot_max]
top = min(int(r["Calls"]) for r in step_kernels)
cnt = {}
for passdir, cname in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    acc = defaultdict(list)
    for r in csv.DictReader(open(one(passdir + "/**/*counter_collection.csv"))):
        if r["Counter_Name"] == cname:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    cnt[cname] = {k: sum(v[len(v) // 4:]) / max(1, len(v[len(v) // 4:])) for k, v in acc.items()}
kern, t_ns, traffic = [], 0.0, 0.0
for r in sorted(step_kernels, key=lambda r: -float(r["TotalDurationNs"])):
    n = r["Name"]
    per_step = int(r["Calls"]) / top
    fb, wb = cnt["FETCH_SIZE"].get(n, 0.0) * 1024 * 2, cnt["WRITE_SIZE"].get(n, 0.0) * 1024
    avg = float(r["AverageNs"])
    kern.append({"kernel": short(n), "calls": int(r["Calls"]), "launches_per_step": per_step, "avg_ns": avg,
                 "ns_per_step": avg * per_step, "fetch_bytes_corrected_x2": fb, "write_bytes": wb,
                 "fetch_bytes_per_step": fb * per_step, "write_bytes_per_step": wb * per_step, "hbm_GBps": (fb + wb) / avg})
    t_ns += avg * per_step
    traffic += (fb + wb) * per_step
alg = BYTES * CELLS
s = {"round": 4, "workload": workload, "steps": top, "kernels_per_step": kern, "step_ns_sum_of_kernel_averages": t_ns,
     "cells_per_step": CELLS, "bytes_per_cell_update": BYTES, "algorithmic_bytes_per_step": alg,
     "traffic_bytes_per_launch": traffic, "traffic_over_algorithmic": traffic / alg,
     "achieved_algorithmic_GBps_under_rocprof": alg / t_ns, "hbm_GBps_under_rocprof": traffic / t_ns,
     "note": "tools/profile_workload.sh + tools/summarize_profile_multi.py: separate rocprofv3 passes (--kernel-trace --stats; "
             "--pmc FETCH_SIZE; --pmc WRITE_SIZE); FETCH_SIZE doubled per MI355X_MICROARCH.md; one 'launch' of this workload is "
             "one step = the kernels listed (launches_per_step > 1: a pass that runs once per chunk of planes, on two streams -- the kernel "
             "times of such passes overlap, their sum exceeds the wall time of a step); averages include the pre-heat launches"}
with open(os.path.join(out, "summary_%s.json" % workload), "w") as fh:
    json.dump(s, fh, indent=1)
with open(os.path.join(out, "bench_%s_kernel_stats.csv" % tag), "w") as fh:
    fh.write(open(one("stats/**/*kernel_stats.csv")).read())
print(json.dumps(s, indent=1))
