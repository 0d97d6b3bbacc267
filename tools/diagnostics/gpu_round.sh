# end-of-milestone run: full GPU test suite, smoke, the bench line of every configuration
set -e
mkdir -p gpurun_out
make -s -C oracle liboracle.so
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r2_pytest.log 2>&1 || (tail -30 gpurun_out/r2_pytest.log; exit 1)
tail -2 gpurun_out/r2_pytest.log
python -c "import __graft_entry__ as g; g.smoke()"
timeout -k 10 400 python bench.py > gpurun_out/bench_C2.json 2> gpurun_out/bench_C2.err || (tail -5 gpurun_out/bench_C2.err; exit 1)
for c in C1 C4 C5; do
timeout -k 10 400 python bench.py --config $c --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/bench_$c.json 2> gpurun_out/bench_$c.err || (tail -5 gpurun_out/bench_$c.err; exit 1)
done
timeout -k 10 400 python bench.py --kernel 2 --steps 60 --no-cpu-baseline > gpurun_out/bench_C2_kernel2.json 2> gpurun_out/bench_C2_k2.err
python - <<'PY'
import json
for n in ("C2","C1","C4","C5","C2_kernel2"):
    d=json.load(open(f"gpurun_out/bench_{n}.json")); c=d.get("compute",{}); r=d["roofline"]
    print(n, round(d["value"],2), "Mpaths/s", round(d["ms_per_step"],3), "ms", "scan launch ms", round(r["avg_launch_ms"],4), "cyc/product", round(r.get("cycles_per_product",0),1), "frac", round(r["frac"],4), "Gtests/s", round(c.get("gtests_per_s",0)), "cand", d["counters_per_frame"]["candidates"], "cpu", d.get("cpu_baseline",{}).get("value"))
PY
