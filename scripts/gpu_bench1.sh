set -e
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()"
python bench.py --steps 5 --warmup 2 > gpurun_out/bench_c2_v0.json 2> gpurun_out/bench_c2_v0.err || (tail -20 gpurun_out/bench_c2_v0.err; exit 1)
cat gpurun_out/bench_c2_v0.json
nproc
