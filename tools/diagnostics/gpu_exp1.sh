set -e
mkdir -p gpurun_out
for g in 8 16 32; do
  timeout -k 10 300 python bench.py --config C4 --steps 20 --warmup 3 --no-cpu-baseline --mf-group-quads $g > gpurun_out/exp_c4_g$g.json 2>/dev/null
  python -c "import json; d=json.load(open('gpurun_out/exp_c4_g$g.json')); print('C4 group', $g, round(d['value'],2), 'Mpaths/s cand', d['counters_per_frame']['candidates'])"
done
for g in 8 16; do
  timeout -k 10 300 python bench.py --config C2 --steps 100 --warmup 5 --no-cpu-baseline --mf-group-quads $g > gpurun_out/exp_c2_g$g.json 2>/dev/null
  python -c "import json; d=json.load(open('gpurun_out/exp_c2_g$g.json')); print('C2 group', $g, round(d['value'],2), 'Mpaths/s cand', d['counters_per_frame']['candidates'])"
done
timeout -k 10 300 python tools/diagnostics/scale_compute_side.py 2>&1 | tee gpurun_out/r2_scale_compute_side.txt
