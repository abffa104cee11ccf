"""Per-kernel table from the three rocprofv3 passes of tools/profile_any.sh: calls, average duration (kernel trace),
FETCH_SIZE x 2 (gfx950 correction for 16-byte streaming reads, MI355X_MICROARCH.md) and WRITE_SIZE per launch, and the
HBM rate they imply.  Usage: python tools/summarize_kernels.py <outdir> [top=8] > table"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 8


def one(pattern):
    f = sorted(glob.glob(os.path.join(out, pattern), recursive=True))
    return f[-1] if f else None


def short(name):
    name = name.replace("void pfhip::(anonymous namespace)::", "").replace("pfhip::(anonymous namespace)::", "")
    return name[:name.index("(")] if "(" in name else name


stats = list(csv.DictReader(open(one("stats/**/*kernel_stats.csv"))))
stats.sort(key=lambda r: -float(r["TotalDurationNs"]))
cnt = {}
for passdir, cname in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    f = one(passdir + "/**/*counter_collection.csv")
    acc = defaultdict(list)
    if f:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == cname:
                acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    cnt[cname] = {k: sum(v[len(v) // 4:]) / max(1, len(v[len(v) // 4:])) for k, v in acc.items()}   # skip the first quarter
rows = []
for r in stats[:top]:
    n = r["Name"]
    fetch = cnt["FETCH_SIZE"].get(n)
    write = cnt["WRITE_SIZE"].get(n)
    avg = float(r["AverageNs"])
    fb = None if fetch is None else fetch * 1024 * 2
    wb = None if write is None else write * 1024
    rows.append({"kernel": short(n), "calls": int(r["Calls"]), "avg_us": avg / 1e3, "pct": float(r["Percentage"]),
                 "fetch_bytes_x2": fb, "write_bytes": wb,
                 "hbm_GBps": None if fb is None or wb is None else (fb + wb) / avg})
print("| kernel | calls | avg us | % | FETCH x2 (MB) | WRITE (MB) | HBM GB/s |")
print("|---|---|---|---|---|---|---|")
for r in rows:
    print("| `%s` | %d | %.1f | %.1f | %s | %s | %s |" % (
        r["kernel"][:100], r["calls"], r["avg_us"], r["pct"],
        "-" if r["fetch_bytes_x2"] is None else "%.1f" % (r["fetch_bytes_x2"] / 1e6),
        "-" if r["write_bytes"] is None else "%.1f" % (r["write_bytes"] / 1e6),
        "-" if r["hbm_GBps"] is None else "%.0f" % r["hbm_GBps"]))
json.dump(rows, open(os.path.join(out, "kernels.json"), "w"), indent=1)
