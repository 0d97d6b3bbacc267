set -e
mkdir -p gpurun_out
rocminfo | grep -m3 -E "Marketing Name|gfx" || true
make -s -C oracle liboracle.so 2>&1 | tail -2
python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -15
