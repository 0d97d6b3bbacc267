set -e
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
OUT=$R/gpurun_out/systrace
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --hip-runtime-trace --output-format csv -d $OUT -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline > $OUT/bench.json 2> $OUT/err.txt || (tail -20 $OUT/err.txt; exit 1)
find $OUT -name "*.csv" | head; du -sh $OUT
