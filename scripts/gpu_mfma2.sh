set -e
mkdir -p gpurun_out
run() {
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" > gpurun_out/ab.json 2> gpurun_out/ab.err || (tail -5 gpurun_out/ab.err; exit 1)
python - "$@" <<'PY'
import json,sys; d=json.load(open('gpurun_out/ab.json')); print(" ".join(sys.argv[1:]), '->', round(d['value'],2),'Mpaths/s', round(d['ms_per_step'],2),'ms', 'avg launch ms', round(d['roofline']['avg_launch_ms'],3), 'scan share', round(d['valu']['scan_share_of_frame'],3))
PY
}
run --kernel 3
run --kernel 3 --debug-skip-exact 1
run --kernel 3 --mf-chunk-quads 16
run --kernel 3 --mf-chunk-quads 16 --debug-skip-exact 1
bash scripts/gpu_profile.sh r1_mfma_c2 --kernel 3
