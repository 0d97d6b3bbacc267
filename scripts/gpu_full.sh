set -e
mkdir -p gpurun_out
make -s -C oracle liboracle.so
python -m pytest tests -m gpu -x -q 2>&1 | tail -5
python bench.py --steps 20 --warmup 3 > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err || (tail -5 gpurun_out/bench_default.err; exit 1)
cat gpurun_out/bench_default.json
for c in C4 C5 C1; do python bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_$c.json 2> gpurun_out/bench_$c.err || (tail -5 gpurun_out/bench_$c.err; exit 1); python -c "
import json; d=json.load(open('gpurun_out/bench_$c.json')); print('$c', round(d['value'],2),'Mpaths/s', round(d['ms_per_step'],2),'ms', round(d['valu']['gtests_per_s']), 'Gtests/s valu frac', round(d['valu']['frac'],3), 'hbm frac', round(d['roofline']['frac'],4))"; done
bash scripts/gpu_profile.sh r1_final_c2
bash scripts/gpu_profile.sh r1_final_c4 --config C4
