set -e
mkdir -p gpurun_out
make -s -C oracle liboracle.so
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -m gpu -x -q 2>&1 | tail -3
timeout -k 10 120 ./tools/mfma_bf16_valu_rate | tee gpurun_out/mfma_bf16_valu_rate.txt
