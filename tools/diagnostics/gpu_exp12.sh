set -e
mkdir -p gpurun_out
make -s -C oracle liboracle.so
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r2_pytest_exp12.log 2>&1 || (tail -40 gpurun_out/r2_pytest_exp12.log; exit 1)
tail -2 gpurun_out/r2_pytest_exp12.log
for d in 1 2; do
  echo "scan_dynamic $d"
  RTGL_AMD_SCAN_DYNAMIC=$d timeout -k 10 120 python tools/diagnostics/solo_frames.py C2 200
  RTGL_AMD_SCAN_DYNAMIC=$d timeout -k 10 120 python tools/diagnostics/solo_frames.py C5 30
  RTGL_AMD_SCAN_DYNAMIC=$d timeout -k 10 120 python tools/diagnostics/solo_frames.py C4 10
done
RTGL_AMD_SCAN_DYNAMIC=1 RTGL_AMD_LIB=$PWD/raytracer.glsl_amd/librtgl_amd_stamps.so timeout -k 10 120 python tools/diagnostics/solo_frames.py C2 10 2>&1 | tee gpurun_out/r2_stamps_C2_static.txt
RTGL_AMD_SCAN_DYNAMIC=2 RTGL_AMD_LIB=$PWD/raytracer.glsl_amd/librtgl_amd_stamps.so timeout -k 10 120 python tools/diagnostics/solo_frames.py C2 10 2>&1 | tee gpurun_out/r2_stamps_C2_dynamic.txt
bash tools/diagnostics/gpu_scaling.sh | tee gpurun_out/r2_scale_compute_side.txt
