set -e
mkdir -p gpurun_out
for i in 1 2; do
( cd _old && python bench.py --steps 10 --warmup 3 --no-cpu-baseline > ../gpurun_out/old.json 2> ../gpurun_out/old.err ) || (tail -5 gpurun_out/old.err; exit 1)
python -c "
import json; d=json.load(open('gpurun_out/old.json')); print('OLD', round(d['value'],2),'Mpaths/s', round(d['ms_per_step'],2),'ms')"
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/new.json 2> gpurun_out/new.err || (tail -5 gpurun_out/new.err; exit 1)
python -c "
import json; d=json.load(open('gpurun_out/new.json')); print('NEW', round(d['value'],2),'Mpaths/s', round(d['ms_per_step'],2),'ms')"
done
