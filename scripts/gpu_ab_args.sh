# usage: bash scripts/gpu_ab_args.sh "<bench args A>" "<bench args B>" ...
set -e
mkdir -p gpurun_out
for a in "$@"; do
python bench.py --steps 20 --warmup 3 --no-cpu-baseline $a > gpurun_out/ab.json 2> gpurun_out/ab.err || (tail -5 gpurun_out/ab.err; exit 1)
python - "$a" <<'PY'
import json,sys; d=json.load(open('gpurun_out/ab.json')); print(sys.argv[1], '->', round(d['value'],2),'Mpaths/s', round(d['ms_per_step'],3),'ms', 'avg launch ms', round(d['roofline']['avg_launch_ms'],3), 'cand', d['counters_per_frame']['candidates'])
PY
done
