#!/bin/bash
# rocprofv3 evidence for one bench.py workload, on the GPU box:   bash tools/profile_workload.sh <workload> <tag> [steps]
# Three separate runs (kernel trace + stats, FETCH_SIZE, WRITE_SIZE -- counters never share a run with traces), then
# tools/summarize_profile.py writes gpurun_out/prof_<tag>/summary_<workload>.json and trimmed CSVs to copy to profiles/.
set -e
W=$1; TAG=$2; STEPS=${3:-40}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="$ROOT/bench.py --workload $W --steps $STEPS --warmup 10 --preheat-s 0.2 --no-cpu-baseline --no-also"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ARGS > $OUT/stats.log 2>&1
echo "stats pass done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $ARGS > $OUT/fetch.log 2>&1
echo "FETCH_SIZE pass done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $ARGS > $OUT/write.log 2>&1
echo "WRITE_SIZE pass done"
case "$W" in
  bm1_fd_512c|bm1_fd_1024c) python3 $ROOT/tools/summarize_profile.py $OUT $W $TAG ;;
  *) python3 $ROOT/tools/summarize_profile_multi.py $OUT $W $TAG ;;
esac
find $OUT -name "*counter_collection.csv" -size +2M -delete 2>/dev/null || true
find $OUT -name "*kernel_trace.csv" -size +2M -delete 2>/dev/null || true
