# how evenly does the static striding spread the scan over the waves?  mean and slowest wave per bounce (10 ns units) against the launch time
set -e
mkdir -p gpurun_out
KERNEL_TIMING=1 RTGL_AMD_LIB=$PWD/raytracer.glsl_amd/librtgl_amd_stamps2.so timeout -k 10 120 python tools/diagnostics/solo_frames.py C2 40 2>&1 | tee gpurun_out/r2_wave_balance_C2.txt
KERNEL_TIMING=1 RTGL_AMD_LIB=$PWD/raytracer.glsl_amd/librtgl_amd_stamps2.so timeout -k 10 120 python tools/diagnostics/solo_frames.py C4 6 2>&1 | tee gpurun_out/r2_wave_balance_C4.txt
