# usage: bash scripts/gpu_trace.sh <tag> [bench args...]  -- kernel trace only, prints the per-kernel summary
set -e
TAG=${1:-t}; shift || true
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
OUT=$R/gpurun_out/trace_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" > $OUT/bench.json 2> $OUT/trace.err || (tail -20 $OUT/trace.err; exit 1)
cat $OUT/bench.json | cut -c1-400
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
cut -c1-200 $f
