set -e
mkdir -p gpurun_out
run() {
python bench.py --steps 20 --warmup 3 --no-cpu-baseline "$@" > gpurun_out/ab.json 2> gpurun_out/ab.err || (tail -5 gpurun_out/ab.err; exit 1)
python - "$@" <<'PY'
import json,sys; d=json.load(open('gpurun_out/ab.json')); print(" ".join(sys.argv[1:]), '->', round(d['value'],2),'Mpaths/s', round(d['ms_per_step'],2),'ms', 'avg launch ms', round(d['roofline']['avg_launch_ms'],3))
PY
}
run
run --sync-each-frame
run --no-kernel-timing
run --no-kernel-timing --sync-each-frame
run
run --sync-each-frame
