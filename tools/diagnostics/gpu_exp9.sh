set -e
mkdir -p gpurun_out
make -s -C oracle liboracle.so
timeout -k 10 1100 python -m pytest tests -m gpu -q > gpurun_out/r2_pytest_exp9.log 2>&1 || (grep -E "^FAILED|passed|failed" gpurun_out/r2_pytest_exp9.log; true)
tail -3 gpurun_out/r2_pytest_exp9.log
FRAMES=100 timeout -k 10 600 python tools/diagnostics/soak_determinism.py
timeout -k 10 120 python tools/diagnostics/solo_frames.py C2 200
timeout -k 10 120 python tools/diagnostics/solo_frames.py C5 30
timeout -k 10 120 python tools/diagnostics/solo_frames.py C4 10
bash tools/diagnostics/gpu_scaling.sh
