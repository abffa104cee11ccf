#!/bin/bash
# rocprofv3 kernel-trace statistics of one python command line, on the GPU box:  bash tools/prof_kernels.sh <tag> <script> [args...]
# prints the per-kernel table (calls, total ms, average us) of the five most expensive kernels
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pk_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $ROOT/"$@" > $OUT/run.log 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True))[-1]
for r in list(csv.DictReader(open(f)))[:6]:
    n = r["Name"].replace("void pfhip::(anonymous namespace)::", "")
    n = n[:n.index("(")] if "(" in n else n
    print("%-34s calls %6s total %9.2f ms avg %9.2f us" % (n[:34], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
PY
find $OUT -name "*kernel_trace.csv" -size +1M -delete 2>/dev/null || true
