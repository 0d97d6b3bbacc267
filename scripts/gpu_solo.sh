set -e
mkdir -p gpurun_out
make -s -C oracle liboracle.so
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_mfma_edges.py -m gpu -x -q 2>&1 | tail -3
bash scripts/gpu_ab_args.sh "" "--kernel 4" "--kernel 4 --mf-group-quads 16"
