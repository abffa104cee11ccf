"""pmc_sq_<workload>.json from the three counter passes of tools/profile_sq.sh: per-launch means of the dominant fused
kernel and the derived utilisations (same definitions as profiles/r01/pmc_sq_bm1_fd_512c.json)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out, workload = sys.argv[1], sys.argv[2]
means, kernel = {}, None
for p in ("p1", "p2", "p3"):
    f = sorted(glob.glob(os.path.join(out, p, "**", "*counter_collection.csv"), recursive=True))[-1]
    acc = defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "ch_fd3d_fused_kernel" in r["Kernel_Name"]:
            kernel = r["Kernel_Name"]
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        v = v[len(v) // 3:]                      # skip the first third (pre-heat start)
        means[k] = sum(v) / len(v)
cyc = means["GRBM_GUI_ACTIVE"] / 8.0
short = kernel[kernel.index("ch_fd3d"):kernel.index("(", kernel.index("ch_fd3d"))] if "(" in kernel[kernel.index("ch_fd3d"):] else kernel
d = {"round": 2, "workload": workload, "kernel": short, "per_launch_means": means, "derived": {
    "gpu_cycles_per_launch(GRBM_GUI_ACTIVE/8 XCDs)": cyc,
    "valu_utilisation = SQ_ACTIVE_INST_VALU*4 / (256 CUs * 4 SIMDs * cycles)": means["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * cyc),
    "lds_utilisation = SQ_LDS_IDX_ACTIVE / (256 CUs * cycles)": means["SQ_LDS_IDX_ACTIVE"] / (256 * cyc),
    "lds_bank_conflict_share = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE": means["SQ_LDS_BANK_CONFLICT"] / means["SQ_LDS_IDX_ACTIVE"],
    "wave_time_waiting(SQ_WAIT_ANY/SQ_WAVE_CYCLES)": means["SQ_WAIT_ANY"] / means["SQ_WAVE_CYCLES"],
    "wave_time_issue_stalled(SQ_WAIT_INST_ANY/SQ_WAVE_CYCLES)": means["SQ_WAIT_INST_ANY"] / means["SQ_WAVE_CYCLES"],
    "wave_time_issuing(SQ_ACTIVE_INST_ANY/SQ_WAVE_CYCLES)": means["SQ_ACTIVE_INST_ANY"] / means["SQ_WAVE_CYCLES"]}}
json.dump(d, open(os.path.join(out, "pmc_sq_%s.json" % workload), "w"), indent=1)
print(json.dumps(d["derived"], indent=1), short)
