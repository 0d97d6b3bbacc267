set -e
mkdir -p gpurun_out
make -s -C oracle liboracle.so
python -m pytest tests -m gpu -x -q 2>&1 | tail -15
python bench.py --steps 10 --warmup 3 > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err || (tail -5 gpurun_out/bench_default.err; exit 1)
cat gpurun_out/bench_default.json
