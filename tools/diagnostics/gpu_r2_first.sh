# round 2, first GPU call: parity of the re-scheduled solo scan, its bench line, the pipe-dense MFMA probe
set -e
mkdir -p gpurun_out
make -s -C oracle liboracle.so
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2_pytest.log 2>&1 || (tail -30 gpurun_out/r2_pytest.log; exit 1)
tail -3 gpurun_out/r2_pytest.log
timeout -k 10 300 python bench.py --steps 40 --warmup 5 --no-fast-mode --no-cpu-baseline > gpurun_out/r2_bench_C2.json 2> gpurun_out/r2_bench_C2.err || (tail -5 gpurun_out/r2_bench_C2.err; exit 1)
python - <<'PY'
import json
d=json.load(open("gpurun_out/r2_bench_C2.json")); c=d["compute"]
print("C2", round(d["value"],2), "Mpaths/s", round(d["ms_per_step"],3), "ms; scan launch", round(d["roofline"]["avg_launch_ms"],4), "ms; Gtests/s", round(c["gtests_per_s"]), "cand", d["counters_per_frame"]["candidates"])
PY
timeout -k 10 300 ./tools/mfma_dense_probe 400000 > gpurun_out/r2_mfma_dense_probe.txt 2>&1
tail -60 gpurun_out/r2_mfma_dense_probe.txt
