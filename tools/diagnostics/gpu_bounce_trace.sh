set -e
R=$(pwd); mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
for d in 1 2; do
  rm -rf $R/gpurun_out/trace_d$d
  RTGL_AMD_SCAN_DYNAMIC=$d rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_d$d -- python3 $R/tools/diagnostics/solo_frames.py ${CFG:-C2} ${FRAMES:-30} > $R/gpurun_out/trace_d$d.log 2>&1
  echo "== scan_dynamic $d"; python3 $R/tools/diagnostics/per_bounce_trace.py $R/gpurun_out/trace_d$d ${BOUNCES:-8}
  rm -rf $R/gpurun_out/trace_d$d
done
