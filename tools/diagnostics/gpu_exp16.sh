set -e
mkdir -p gpurun_out
for g in 8 16 32; do for c in 16 32; do
  timeout -k 10 300 python bench.py --config C4 --steps 10 --warmup 3 --no-cpu-baseline --mf-group-quads $g --mf-chunk-quads $c > gpurun_out/exp16.json 2>/dev/null
  python -c "import json; d=json.load(open('gpurun_out/exp16.json')); print('C4 group', $g, 'chunk', $c, round(d['value'],2), 'Mpaths/s candidates', d['counters_per_frame']['candidates'])"
done; done
