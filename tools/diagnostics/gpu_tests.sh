set -e
mkdir -p gpurun_out
make -s -C oracle liboracle.so
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r2_pytest.log 2>&1 || (tail -40 gpurun_out/r2_pytest.log; exit 1)
tail -2 gpurun_out/r2_pytest.log
timeout -k 10 120 python tools/diagnostics/solo_frames.py C2 60
timeout -k 10 120 python tools/diagnostics/solo_frames.py C5 20
