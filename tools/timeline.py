"""Timeline of a rocprofv3 --kernel-trace run: the last N kernel launches with start offset, duration and the gap to the
previous kernel's end (all queues merged, sorted by start).  Usage: python tools/timeline.py <outdir> [last=60]"""
import csv
import glob
import sys

out = sys.argv[1]
last = int(sys.argv[2]) if len(sys.argv) > 2 else 60
f = sorted(glob.glob(out + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"), r.get("Stream_Id", "?"))
        for r in csv.DictReader(open(f))]
rows.sort()
rows = rows[-last:]
t0 = rows[0][0]
prev_end = rows[0][0]
print("%10s %9s %9s  q/stream  kernel" % ("start us", "dur us", "gap us"))
for s, e, name, q, st in rows:
    print("%10.1f %9.1f %9.1f  %s/%s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, q, st, name[:70]))
    prev_end = max(prev_end, e)
print("window %.1f us" % ((max(r[1] for r in rows) - t0) / 1e3))
