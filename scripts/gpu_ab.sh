set -e
mkdir -p gpurun_out
run() {
python bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" > gpurun_out/ab.json 2> gpurun_out/ab.err || (tail -5 gpurun_out/ab.err; exit 1)
python - "$@" <<'PY'
import json,sys; d=json.load(open('gpurun_out/ab.json')); print(" ".join(sys.argv[1:]), '->', round(d['value'],2),'Mpaths/s', round(d['ms_per_step'],2),'ms', 'avg launch ms', round(d['roofline']['avg_launch_ms'],3))
PY
}
for i in 1 2; do
run --wf-packed 1 --wf-rays 4
run --wf-packed 0 --wf-rays 4
run --wf-packed 1 --wf-rays 2
run --wf-packed 0 --wf-rays 2
run --wf-packed 1 --wf-rays 8
run --wf-packed 0 --wf-rays 8
done
run --wf-packed 0 --wf-rays 4 --wf-chunk 256
run --wf-packed 0 --wf-rays 4 --wf-chunk 1024
