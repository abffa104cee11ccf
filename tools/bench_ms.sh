#!/bin/bash
# ms/step and roofline fraction of bench workloads, N fresh processes each: tools/bench_ms.sh "workloads" [N]
for w in $1; do for p in $(seq 1 ${2:-2}); do
  printf "%-20s process %d: " $w $p
  python bench.py --workload $w --steps 20 --warmup 5 --no-also --no-cpu-baseline 2>/dev/null | python -c '
import json,sys
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); print("%.4f ms/step  frac %s" % (d["ms_per_step"], (d.get("roofline") or {}).get("frac")))
'
done; done
