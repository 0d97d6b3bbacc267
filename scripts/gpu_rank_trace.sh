set -e
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
OUT=$R/gpurun_out/ranktrace; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $R && rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $R/scripts/dbg_rank_trace.py > $OUT/log.txt 2>&1 || (tail $OUT/log.txt; exit 1)
