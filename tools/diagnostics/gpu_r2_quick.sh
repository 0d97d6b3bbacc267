# quick loop: parity subset, C2 frame time, cycle stamps
set -e
mkdir -p gpurun_out
make -s -C oracle liboracle.so
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_mfma_edges.py tests/test_gpu_fullsize.py -m gpu -x -q > gpurun_out/r2_pytest.log 2>&1 || (tail -30 gpurun_out/r2_pytest.log; exit 1)
tail -2 gpurun_out/r2_pytest.log
timeout -k 10 120 python tools/diagnostics/solo_frames.py C2 30
RTGL_AMD_LIB=$PWD/raytracer.glsl_amd/librtgl_amd_stamps.so timeout -k 10 120 python tools/diagnostics/solo_frames.py C2 10 2>&1 | tee gpurun_out/r2_stamps_C2.txt
