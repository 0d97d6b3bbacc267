set -e
mkdir -p gpurun_out
make -s -C oracle liboracle.so
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r2_pytest.log 2>&1 || (tail -40 gpurun_out/r2_pytest.log; exit 1)
tail -2 gpurun_out/r2_pytest.log
python tools/diagnostics/rank_frames.py 1 0 100
python tools/diagnostics/rank_frames.py 2 1 100
python tools/diagnostics/rank_frames.py 4 3 100
python tools/diagnostics/rank_frames.py 8 0 100
python tools/diagnostics/rank_frames.py 8 7 100
