#!/bin/bash
# rocprofv3 evidence for any bench.py command line, on the GPU box:  bash tools/profile_any.sh <tag> <bench.py args...>
# Three separate runs (kernel trace + stats, FETCH_SIZE, WRITE_SIZE: counters never share a run with traces), then
# tools/summarize_kernels.py prints the per-kernel table into gpurun_out/prof_<tag>/kernels.md
set -e
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/bench.py "$@" > $OUT/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $ROOT/bench.py "$@" > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $ROOT/bench.py "$@" > $OUT/write.log 2>&1
python3 $ROOT/tools/summarize_kernels.py $OUT 10 > $OUT/kernels.md
cat $OUT/kernels.md
