set -e
mkdir -p gpurun_out
make -s -C oracle liboracle.so
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r2_pytest_exp14.log 2>&1 || (tail -40 gpurun_out/r2_pytest_exp14.log; exit 1)
tail -2 gpurun_out/r2_pytest_exp14.log
timeout -k 10 120 python tools/diagnostics/solo_frames.py C2 300
timeout -k 10 120 python tools/diagnostics/solo_frames.py C5 40
timeout -k 10 120 python tools/diagnostics/solo_frames.py C4 12
RTGL_AMD_SCAN_DYNAMIC=1 timeout -k 10 120 python tools/diagnostics/solo_frames.py C4 12
timeout -k 10 120 python tools/diagnostics/solo_frames.py C1 300
bash tools/diagnostics/gpu_scaling.sh | tee gpurun_out/r2_scale_compute_side.txt
N=60 timeout -k 10 400 python tools/diagnostics/flaky_multi.py | grep "differing images"
