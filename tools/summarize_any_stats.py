"""Kernel-time breakdown of a rocprofv3 --kernel-trace --stats run: share, calls, average duration per kernel.
Usage: python tools/summarize_any_stats.py <outdir> [top=25] [kernel-substring: its launches grouped by grid size]"""
import csv
import glob
import sys

out = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
f = sorted(glob.glob(out + "/**/*kernel_stats.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time %.1f ms (sum over all streams)" % (tot / 1e6))
for r in rows[:top]:
    print("%6.1f%% %8d calls %9.1f us avg  %s" % (100 * float(r["TotalDurationNs"]) / tot, int(r["Calls"]),
                                                float(r["AverageNs"]) / 1e3, r["Name"][:110]))

if len(sys.argv) > 3:
    key = sys.argv[3]
    tr = sorted(glob.glob(out + "/**/*kernel_trace.csv", recursive=True))
    if tr:
        groups = {}
        for r in csv.DictReader(open(tr[-1])):
            if key in r["Kernel_Name"]:
                g = (int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1), int(r["Grid_Size_Y"]) // max(int(r["Workgroup_Size_Y"]), 1))
                groups.setdefault(g, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        seq = sorted((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
                     for r in csv.DictReader(open(tr[-1])) if key in r["Kernel_Name"])
        print("first launches of *%s* in order, us: %s" % (key, " ".join("%.0f" % d for _, d in seq[:21])))
        print("launches of *%s* by grid (workgroups x, y): calls, average us" % key)
        for g in sorted(groups, key=lambda t: -t[1]):
            v = groups[g]
            print("  grid %4d x %3d  %6d calls  %9.1f us" % (g[0], g[1], len(v), sum(v) / len(v)))
