# quick acceptance of a scan-kernel change: parity subset, survivor-set determinism, throughput
set -e
mkdir -p gpurun_out
make -s -C oracle liboracle.so
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_mfma_edges.py -m gpu -x -q 2>&1 | tail -2
python scripts/dbg_cand.py 2>&1 | tail -3 | cut -c1-200
FRAMES=12 python scripts/dbg_soak.py 2>&1 | tail -3
bash scripts/gpu_ab_args.sh "" "--mf-group-quads 32" "--kernel 2"
