#!/bin/bash
# every bench workload under the allocation policies of csrc/device_alloc.hip: tools/alloc_ab.sh "policies" "workloads"
for w in $2; do for pol in $1; do
  printf "%-22s PFHIP_ALLOC=%-14s " $w $pol
  PFHIP_ALLOC=$pol python bench.py --workload $w --steps ${STEPS:-40} --warmup 10 --no-also --no-cpu-baseline 2>/dev/null | python -c '
import json,sys
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); r=d.get("roofline") or {}
        print("%.4f ms/step  value %.4g %s  frac %s" % (d["ms_per_step"], d["value"], d["unit"], r.get("frac")))
'
done; done
