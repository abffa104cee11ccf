#!/bin/bash
# round-end evidence on one box: the driver's bench command, fresh processes of the placement-sensitive workloads, rocprofv3 summaries
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_driverflags_r4.log 2> gpurun_out/bench_driverflags_r4.err
echo "bench done"
for w in bm1_spectral_512c bm3_fd_512c bm6_fd_512c bm2_fd_512c; do for p in 1 2 3; do
  printf "%-20s process %d: " $w $p
  python bench.py --workload $w --steps 20 --warmup 5 --no-also --no-cpu-baseline 2>/dev/null | python -c '
import json,sys
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); print("%.4f ms/step  frac %s" % (d["ms_per_step"], (d.get("roofline") or {}).get("frac")))
'
done; done > gpurun_out/three_processes_r4.log 2>&1
echo "three processes done"
for w in bm1_spectral_512c bm3_fd_512c bm2_fd_512c bm6_fd_512c; do
  bash tools/profile_workload.sh $w r4f_$w 40 > gpurun_out/prof_r4f_$w.log 2>&1 || echo "profile of $w failed"
  echo "profile $w done"
done
