# the bench line of C2 / C4 / C5 (short), no tests
set -e
mkdir -p gpurun_out
timeout -k 10 300 python bench.py --steps 200 --warmup 10 --no-cpu-baseline > gpurun_out/r3_bench_C2.json 2> gpurun_out/r3_bench_C2.err || (tail -5 gpurun_out/r3_bench_C2.err; exit 1)
for c in C4 C5; do
timeout -k 10 300 python bench.py --config $c --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r3_bench_$c.json 2> gpurun_out/r3_bench_$c.err || (tail -5 gpurun_out/r3_bench_$c.err; exit 1)
done
python - <<'PY'
import json
for n in ("C2","C4","C5"):
    d=json.load(open(f"gpurun_out/r3_bench_{n}.json")); r=d["roofline"]; cn=d["counters_per_frame"]
    print(n, round(d["value"],2), "Mpaths/s", round(d["ms_per_step"],3), "ms", "scan launch ms", round(r["avg_launch_ms"],4), "frac (executed)", round(r["frac"],4), "culled", round(cn.get("culled_tests",0)/max(cn["triangle_tests"],1),4), "cand", cn["candidates"], "batched", d.get("frame_batched",{}).get("value"))
PY
