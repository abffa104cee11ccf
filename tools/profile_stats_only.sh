#!/bin/bash
# Kernel-time breakdown (rocprofv3 --kernel-trace --stats) of one bench.py command line, on the GPU box:
#   bash tools/profile_stats_only.sh <tag> <bench.py args...>   ->  gpurun_out/stats_<tag>/breakdown.txt
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/stats_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/bench.py "$@" > $OUT/stats.log 2>&1
rc=$?
cd $ROOT
python3 tools/summarize_any_stats.py $OUT/stats 30 ${PF_GROUP_KERNEL:-lu_solve} > $OUT/breakdown.txt
rm -f $OUT/stats/*/*kernel_trace.csv $OUT/stats/*/*_agent_info.csv
exit $rc
