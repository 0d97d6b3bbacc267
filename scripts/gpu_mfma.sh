set -e
mkdir -p gpurun_out
make -s -C oracle liboracle.so
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -m gpu -x -q 2>&1 | tail -3
run() {
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" > gpurun_out/ab.json 2> gpurun_out/ab.err || (tail -5 gpurun_out/ab.err; exit 1)
python - "$@" <<'PY'
import json,sys; d=json.load(open('gpurun_out/ab.json')); print(" ".join(sys.argv[1:]), '->', round(d['value'],2),'Mpaths/s', round(d['ms_per_step'],2),'ms', 'avg launch ms', round(d['roofline']['avg_launch_ms'],3), 'cand', d['counters_per_frame']['candidates'])
PY
}
run --kernel 3 --mf-group-quads 4
run --kernel 3 --mf-group-quads 8
run --kernel 3 --mf-group-quads 16
run --kernel 3 --mf-group-quads 32
run --kernel 3 --mf-group-quads 64
run --kernel 3 --mf-group-quads 16 --mf-chunk-quads 32
run --kernel 3 --mf-group-quads 32 --mf-chunk-quads 128
