# kernel trace of one rank of WORLD (compute side only, no gather): per-bounce scan durations and the other kernels per frame
set -e
R=$(pwd); mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/trace_rank
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_rank -- python3 $R/tools/diagnostics/rank_frames.py ${WORLD:-8} 0 40 > $R/gpurun_out/trace_rank.log 2>&1
tail -1 $R/gpurun_out/trace_rank.log
python3 $R/tools/diagnostics/per_bounce_trace.py $R/gpurun_out/trace_rank 8 ${TIMELINE:+timeline}
rm -rf $R/gpurun_out/trace_rank
