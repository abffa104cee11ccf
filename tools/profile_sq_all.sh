#!/bin/bash
# SQ / GRBM counter passes (counters only, never with a trace) for EVERY kernel of a bench.py workload:
#   bash tools/profile_sq_all.sh <workload> <tag>   ->  gpurun_out/prof_sq_<tag>/sq_kernels.{md,json}
set -e
W=$1; TAG=$2
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_sq_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="$ROOT/bench.py --workload $W --steps 10 --warmup 2 --preheat-s 0.1 --repeats 1 --no-cpu-baseline --no-also"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY --output-format csv -d $OUT/p1 -- python3 $ARGS > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_WAVES --output-format csv -d $OUT/p2 -- python3 $ARGS > $OUT/p2.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $OUT/p3 -- python3 $ARGS > $OUT/p3.log 2>&1
python3 $ROOT/tools/summarize_sq_all.py $OUT > $OUT/sq_kernels.md
find $OUT -name "*counter_collection.csv" -size +4M -delete 2>/dev/null || true
cat $OUT/sq_kernels.md
