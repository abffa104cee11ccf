#!/bin/bash
# A/B of kernel forms of the 512^3 spectral step on the GPU box: one rocprofv3 --kernel-trace --stats run of
# `bench.py --workload <W>` per environment variant, per-kernel average durations into gpurun_out/variants_<tag>/table.md.
#   bash tools/spectral_variants.sh <tag> <workload> "VAR=1 VAR2=x" "VAR=2" ...      ("-" = no variables)
# A variant prefixed with "pmc:" additionally gets FETCH_SIZE and WRITE_SIZE passes (separate runs, counters only).
set -e
TAG=$1; W=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/variants_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="$ROOT/bench.py --workload $W --steps 20 --warmup 5 --preheat-s 0.3 --no-cpu-baseline --no-also"
i=0
: > $OUT/table.md
for V in "$@"; do
  i=$((i+1))
  PMC=0
  case "$V" in pmc:*) PMC=1; V=${V#pmc:};; esac
  D=$OUT/v$i
  mkdir -p $D
  echo "## variant $i: $V" >> $OUT/table.md
  ( [ "$V" != "-" ] && export $V; rocprofv3 --kernel-trace --stats --output-format csv -d $D/stats -- python3 $ARGS > $D/stats.log 2>&1 )
  if [ $PMC = 1 ]; then
    ( [ "$V" != "-" ] && export $V; rocprofv3 --pmc FETCH_SIZE --output-format csv -d $D/fetch -- python3 $ARGS > $D/fetch.log 2>&1 )
    ( [ "$V" != "-" ] && export $V; rocprofv3 --pmc WRITE_SIZE --output-format csv -d $D/write -- python3 $ARGS > $D/write.log 2>&1 )
  fi
  grep -o '"ms_per_step": [0-9.]*' $D/stats.log | head -1 >> $OUT/table.md
  grep -o '"F_after": [0-9.e+-]*' $D/stats.log | head -1 >> $OUT/table.md
  python3 $ROOT/tools/summarize_kernels.py $D 6 >> $OUT/table.md
  find $D -name "*.csv" ! -name "*kernel_stats.csv" -delete 2>/dev/null || true
  echo "variant $i done: $V"
done
cat $OUT/table.md
