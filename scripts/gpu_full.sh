set -e
mkdir -p gpurun_out
make -s -C oracle liboracle.so
timeout -k 10 1100 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')"
timeout -k 10 400 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err || (tail -5 gpurun_out/bench_default.err; exit 1)
cat gpurun_out/bench_default.json
