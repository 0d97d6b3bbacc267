set -e
mkdir -p gpurun_out
make -s -C oracle liboracle.so
python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -15
run() {
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --kernel $1 --wf-mode $2 --wf-rays $3 --wf-chunk $4 > gpurun_out/var_$1$2$3_$4.json 2> gpurun_out/var_$1$2$3_$4.err || (tail -5 gpurun_out/var_$1$2$3_$4.err; exit 1)
  python - <<PY
import json; d=json.load(open("gpurun_out/var_$1$2$3_$4.json")); print("kernel/mode/rays/chunk $1 $2 $3 $4:", round(d["value"],2), "Mpaths/s", round(d["ms_per_step"],2), "ms", round(d["valu"]["gtests_per_s"],1), "Gtests/s")
PY
}
for v in "1 1 2 1024" "2 0 1 1024" "2 0 2 1024" "2 0 4 1024" "2 0 8 1024" "2 1 1 1024" "2 1 2 1024" "2 1 4 1024" "2 1 8 1024" "2 1 4 512" "2 1 4 2560" "2 0 4 512" "2 0 4 2560" "2 1 8 2560"; do run $v; done
