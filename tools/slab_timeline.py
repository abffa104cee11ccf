"""Timeline of the slab (multi-GPU) step on one GPU from a rocprofv3 --kernel-trace CSV: per step the interior launch,
the exchange kernel(s) and the boundary launch with start offsets, durations and gaps.
Usage: python tools/slab_timeline.py <kernel_trace.csv> [first_step=30] [nsteps=4]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
first = int(sys.argv[2]) if len(sys.argv) > 2 else 30
nsteps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
fused = [i for i, r in enumerate(rows) if "ch_fd3d_fused_kernel" in r["Kernel_Name"]]
# a step = interior launch (long) followed by the boundary launch (short): find long launches
durs = {i: int(rows[i]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"]) for i in fused}
longs = [i for i in fused if durs[i] > 0.5 * max(durs.values())]
t0 = int(rows[longs[first]]["Start_Timestamp"])
lo, hi = longs[first], longs[first + nsteps]
for r in rows[lo:hi]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"]
    name = name[:name.index("(")] if "(" in name else name
    print("%9.1f us  +%8.1f us  q=%s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), name[:90]))
per = (int(rows[longs[first + nsteps]]["Start_Timestamp"]) - t0) / nsteps / 1e3
print("step period over %d steps: %.1f us" % (nsteps, per))
