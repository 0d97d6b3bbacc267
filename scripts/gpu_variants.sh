set -e
mkdir -p gpurun_out
make -s -C oracle liboracle.so
python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -15
for v in "0 0 1" "1 0 1" "1 0 2" "1 0 4" "1 1 1" "1 1 2" "1 1 4"; do
  set -- $v
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --kernel $1 --wf-mode $2 --wf-rays $3 > gpurun_out/var_$1$2$3.json 2> gpurun_out/var_$1$2$3.err || (tail -5 gpurun_out/var_$1$2$3.err; exit 1)
  python - <<PY
import json; d=json.load(open("gpurun_out/var_$1$2$3.json")); print("kernel/mode/rays $1 $2 $3:", round(d["value"],2), "Mpaths/s", round(d["ms_per_step"],2), "ms", round(d["valu"]["gtests_per_s"],1), "Gtests/s")
PY
done
