# end of round: full suite, smoke, bench lines, profiles of C2 / C4 / C5, soak, scaling (compute side), stamps
set -e
mkdir -p gpurun_out
bash tools/diagnostics/gpu_round.sh
# the whole suite again with the scan's other forms as the default of every context that does not choose for itself
for e in "RTGL_AMD_SCAN_WAVES=1" "RTGL_AMD_SCAN_DYNAMIC=2" "RTGL_AMD_SCAN_DYNAMIC=2 RTGL_AMD_SCAN_WAVES=1" "RTGL_AMD_FRAME_BATCH=4"; do echo "== $e"; env $e timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -1; done
bash tools/diagnostics/gpu_profiles_all.sh > gpurun_out/profiles_all.log 2>&1 || (tail -20 gpurun_out/profiles_all.log; exit 1)
FRAMES=300 timeout -k 10 600 python tools/diagnostics/soak_determinism.py | tee gpurun_out/r2_soak_c2.txt
CFG=C5 FRAMES=12 timeout -k 10 300 python tools/diagnostics/soak_determinism.py | tee gpurun_out/r2_soak_c5.txt
CFG=C4 FRAMES=6 timeout -k 10 300 python tools/diagnostics/soak_determinism.py | tee gpurun_out/r2_soak_c4.txt
bash tools/diagnostics/gpu_scaling.sh | tee gpurun_out/r2_scale_compute_side.txt
RTGL_AMD_LIB=$PWD/raytracer.glsl_amd/librtgl_amd_stamps.so timeout -k 10 120 python tools/diagnostics/solo_frames.py C2 10 2>&1 | tee gpurun_out/r2_stamps_C2.txt
RTGL_AMD_LIB=$PWD/raytracer.glsl_amd/librtgl_amd_stamps.so timeout -k 10 120 python tools/diagnostics/solo_frames.py C4 4 2>&1 | tee gpurun_out/r2_stamps_C4.txt
