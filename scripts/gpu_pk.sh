set -e
mkdir -p gpurun_out
make -s -C oracle liboracle.so
python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -m gpu -x -q 2>&1 | tail -4
run() {
  python bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" > gpurun_out/tmp.json 2> gpurun_out/tmp.err || (tail -5 gpurun_out/tmp.err; exit 1)
  python - "$@" <<'PY'
import json, sys; d=json.load(open("gpurun_out/tmp.json")); print(" ".join(sys.argv[1:]), "->", round(d["value"],2), "Mpaths/s", round(d["ms_per_step"],2), "ms", round(d["valu"]["gtests_per_s"]), "Gtests/s")
PY
}
run --wf-rays 2
run --wf-rays 4
run --wf-rays 8
run --wf-rays 4 --wf-chunk 256
run --wf-rays 8 --wf-chunk 256
run --wf-mode 0 --wf-rays 4
run --wf-mode 0 --wf-rays 8
run --wf-rays 1
