# dynamic work distribution of the scan: full GPU suite, frame times, wave balance, scaling
set -e
mkdir -p gpurun_out
make -s -C oracle liboracle.so
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r2_pytest_exp8.log 2>&1 || (tail -40 gpurun_out/r2_pytest_exp8.log; exit 1)
tail -2 gpurun_out/r2_pytest_exp8.log
timeout -k 10 120 python tools/diagnostics/solo_frames.py C2 200
timeout -k 10 120 python tools/diagnostics/solo_frames.py C5 30
timeout -k 10 120 python tools/diagnostics/solo_frames.py C4 10
KERNEL_TIMING=1 RTGL_AMD_LIB=$PWD/raytracer.glsl_amd/librtgl_amd_stamps2.so timeout -k 10 120 python tools/diagnostics/solo_frames.py C2 40 2>&1 | tee gpurun_out/r2_wave_balance_C2.txt
KERNEL_TIMING=1 RTGL_AMD_LIB=$PWD/raytracer.glsl_amd/librtgl_amd_stamps2.so timeout -k 10 120 python tools/diagnostics/solo_frames.py C4 6 2>&1 | tee gpurun_out/r2_wave_balance_C4.txt
RTGL_AMD_LIB=$PWD/raytracer.glsl_amd/librtgl_amd_stamps.so timeout -k 10 120 python tools/diagnostics/solo_frames.py C2 10 2>&1 | tee gpurun_out/r2_stamps_C2_auto.txt
bash tools/diagnostics/gpu_scaling.sh | tee gpurun_out/r2_scale_compute_side.txt
