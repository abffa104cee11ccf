#!/bin/bash
# Whole-process wall time of the BE-parity drivers, with and without the second stream of the dense reduction levels.
# Usage on the GPU box: bash tools/time_drivers.sh "1 2 3" 2     (benchmarks, repetitions)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for b in ${1:-1 2}; do
  for rep in $(seq 1 ${2:-1}); do
    for st in 0 1; do
      export PFHIP_FEM_STREAMS=$st
      t0=$(date +%s%N)
      python3 $ROOT/dolfin/bench$b.py --quiet --out-dir /tmp/res_td$b > /dev/null 2>&1 || echo "bench$b failed"
      t1=$(date +%s%N)
      echo "bench$b PFHIP_FEM_STREAMS=$st: $(( (t1 - t0) / 1000000 )) ms"
    done
  done
done
