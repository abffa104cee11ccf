"""Per-handle kernel averages from ONE rocprofv3 kernel trace of a process that creates several handles in turn
(tools/spectral_env_ab.py): the trace is cut at the memset launches of pf_create, and for every segment the average
duration of each spectral pass is printed -- which passes differ between a 'fast' and a 'slow' handle of one process?
Usage: python tools/segment_kernels.py <dir with *kernel_trace.csv>"""
import csv
import glob
import sys
from collections import defaultdict

f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
segs, cur = [], defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    if "fillBuffer" in n:
        if sum(len(v) for v in cur.values()) > 50:
            segs.append(cur)
            cur = defaultdict(list)
        continue
    if "f3_" in n or "f2_row512" in n:
        k = n[:n.index("(")] if "(" in n else n
        k = k.replace("void ", "").replace("pfhip::", "").replace("(anonymous namespace)::", "")
        k += "/%s" % r.get("Grid_Size_X", r.get("Grid_Size", "?"))     # the z pass and the y passes share one kernel template
        cur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
if cur:
    segs.append(cur)
for i, s in enumerate(segs):
    parts = []
    tot = 0.0
    for k in sorted(s):
        v = s[k][len(s[k]) // 3:]          # skip the first third (pre-heat, power transient)
        if len(v) < 5:
            continue
        a = sum(v) / len(v)
        tot += a
        parts.append("%s %.1f" % (k.replace("_kernel", ""), a))
    print("handle %2d: sum %.1f us | %s" % (i, tot, " | ".join(parts)))
