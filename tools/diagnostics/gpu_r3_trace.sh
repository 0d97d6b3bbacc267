# per-bounce kernel trace of a configuration for a list of option sets: CFG=C2 FRAMES=30 SETS="cull=1 cull=3" bash tools/diagnostics/gpu_r3_trace.sh
set -e
R=$(pwd); mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
for o in ${SETS:-cull=1 cull=3}; do
  rm -rf $R/gpurun_out/trace_tmp
  OPTS=$o rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_tmp -- python3 $R/tools/diagnostics/solo_frames.py ${CFG:-C2} ${FRAMES:-30} > $R/gpurun_out/trace_$o.log 2>&1
  echo "== $o: $(grep Mpaths $R/gpurun_out/trace_$o.log)"; python3 $R/tools/diagnostics/per_bounce_trace.py $R/gpurun_out/trace_tmp ${BOUNCES:-8}
  rm -rf $R/gpurun_out/trace_tmp
done
