# the timeline of a frame of one rank's strips (WORLD, RANK on one GPU): every kernel with its duration and the gap before it
set -e
R=$(pwd); mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/trace_tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_tmp -- python3 $R/tools/diagnostics/rank_sweep.py ${WORLD:-8} ${RANK:-0} ${OPTS:-} > $R/gpurun_out/rank_timeline.log 2>&1
python3 $R/tools/diagnostics/per_bounce_trace.py $R/gpurun_out/trace_tmp 8 timeline > $R/gpurun_out/rank_timeline.txt
rm -rf $R/gpurun_out/trace_tmp
cat $R/gpurun_out/rank_timeline.txt
