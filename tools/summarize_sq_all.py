"""Per-kernel SQ / GRBM counter table from the three passes of tools/profile_sq_all.sh (counters only, no traces): mean
per launch of every kernel that accounts for >= 2 % of SQ_WAVE_CYCLES, and the derived utilisations.
Usage: python tools/summarize_sq_all.py <outdir> > table.md"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out = sys.argv[1]
means = defaultdict(dict)
for p in ("p1", "p2", "p3"):
    fs = sorted(glob.glob(os.path.join(out, p, "**", "*counter_collection.csv"), recursive=True))
    if not fs:
        continue
    acc = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(fs[-1])):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        for c, v in d.items():
            v = v[len(v) // 3:] if len(v) >= 6 else v
            means[k][c] = sum(v) / len(v)
            means[k]["_n_" + c] = len(v)


def short(name):
    name = name.replace("void pfhip::(anonymous namespace)::", "").replace("pfhip::(anonymous namespace)::", "")
    return name[:name.index("(")] if "(" in name else name


tot = sum(m.get("SQ_WAVE_CYCLES", 0) * m.get("_n_SQ_WAVE_CYCLES", 0) for m in means.values()) or 1.0
rows = []
for k, m in means.items():
    if "SQ_WAVE_CYCLES" not in m or m["SQ_WAVE_CYCLES"] * m["_n_SQ_WAVE_CYCLES"] < 0.02 * tot:
        continue
    cyc = m.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    wc = m["SQ_WAVE_CYCLES"]
    g = lambda c: m.get(c, float("nan"))
    rows.append({
        "kernel": short(k), "gpu_cycles": cyc, "waves": g("SQ_WAVES"),
        "mean_waves_per_cu = SQ_WAVE_CYCLES*4/(256*cycles)": wc * 4 / (256 * cyc) if cyc else None,
        "valu_util": g("SQ_ACTIVE_INST_VALU") * 4 / (1024 * cyc) if cyc else None,
        "lds_util": g("SQ_LDS_IDX_ACTIVE") / (256 * cyc) if cyc else None,
        "lds_conflict_share": g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE") if g("SQ_LDS_IDX_ACTIVE") else None,
        "wait_any": g("SQ_WAIT_ANY") / wc, "wait_inst": g("SQ_WAIT_INST_ANY") / wc, "issuing": g("SQ_ACTIVE_INST_ANY") / wc,
        "insts_valu": g("SQ_INSTS_VALU"), "insts_lds": g("SQ_INSTS_LDS"), "insts_vmem_rd": g("SQ_INSTS_VMEM_RD"),
        "insts_vmem_wr": g("SQ_INSTS_VMEM_WR"), "insts_salu": g("SQ_INSTS_SALU")})
rows.sort(key=lambda r: -r["gpu_cycles"])
print("| kernel | gpu cycles | waves | waves/CU | VALU | LDS | LDS confl | waiting | issue-stalled | issuing |")
print("|---|---|---|---|---|---|---|---|---|---|")
f = lambda v: "-" if v is None or v != v else "%.2f" % v
for r in rows:
    print("| `%s` | %.0f | %.0f | %s | %s | %s | %s | %s | %s | %s |" % (
        r["kernel"][:80], r["gpu_cycles"], r["waves"], f(r["mean_waves_per_cu = SQ_WAVE_CYCLES*4/(256*cycles)"]),
        f(r["valu_util"]), f(r["lds_util"]), f(r["lds_conflict_share"]), f(r["wait_any"]), f(r["wait_inst"]), f(r["issuing"])))
json.dump(rows, open(os.path.join(out, "sq_kernels.json"), "w"), indent=1)
