# after moving the certificate into its own kernel: full GPU suite, frame times, cycle stamps (with the slowest wave per bounce)
set -e
mkdir -p gpurun_out
make -s -C oracle liboracle.so
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r2_pytest_exp6.log 2>&1 || (tail -40 gpurun_out/r2_pytest_exp6.log; exit 1)
tail -2 gpurun_out/r2_pytest_exp6.log
timeout -k 10 120 python tools/diagnostics/solo_frames.py C2 200
timeout -k 10 120 python tools/diagnostics/solo_frames.py C5 30
timeout -k 10 120 python tools/diagnostics/solo_frames.py C4 10
RTGL_AMD_LIB=$PWD/raytracer.glsl_amd/librtgl_amd_stamps.so timeout -k 10 120 python tools/diagnostics/solo_frames.py C2 10 2>&1 | tee gpurun_out/r2_stamps_C2_auto.txt
python tools/diagnostics/rank_frames.py 8 0 100
