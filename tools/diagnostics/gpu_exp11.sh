set -e
R=$(pwd); mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/trace_new $R/gpurun_out/trace_prev
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_new -- python3 $R/tools/diagnostics/solo_frames.py C2 30 > $R/gpurun_out/trace_new.log 2>&1
RTGL_AMD_LIB=$R/raytracer.glsl_amd/librtgl_amd_prev.so rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_prev -- python3 $R/tools/diagnostics/solo_frames.py C2 30 > $R/gpurun_out/trace_prev.log 2>&1
cd $R
echo "== dynamic (this build)"; python tools/diagnostics/per_bounce_trace.py gpurun_out/trace_new 8
echo "== static (previous commit)"; python tools/diagnostics/per_bounce_trace.py gpurun_out/trace_prev 8
rm -rf gpurun_out/trace_new gpurun_out/trace_prev
