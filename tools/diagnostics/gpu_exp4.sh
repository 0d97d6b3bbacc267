set -e
for c in 0 1 2; do
  timeout -k 10 300 python bench.py --config C2 --steps 100 --warmup 5 --no-cpu-baseline --cull $c > gpurun_out/exp_cull$c.json 2>/dev/null
  python -c "import json; d=json.load(open('gpurun_out/exp_cull$c.json')); print('C2 cull', $c, round(d['value'],1), 'Mpaths/s culled', round(d['roofline']['executed']['culled_fraction'],4))"
done
