# the timeline of a frame (every kernel with its duration and the gap before it): CFG=C2 FRAMES=30 [OPTS=...]
set -e
R=$(pwd); mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/trace_tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_tmp -- python3 $R/tools/diagnostics/solo_frames.py ${CFG:-C2} ${FRAMES:-30} > $R/gpurun_out/timeline.log 2>&1
python3 $R/tools/diagnostics/per_bounce_trace.py $R/gpurun_out/trace_tmp ${BOUNCES:-8} timeline > $R/gpurun_out/timeline_${CFG:-C2}.txt
rm -rf $R/gpurun_out/trace_tmp
cat $R/gpurun_out/timeline_${CFG:-C2}.txt
