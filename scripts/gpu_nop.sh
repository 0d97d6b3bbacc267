set -e
for v in nop0 nop3 nop7 nop15; do
export RTGL_AMD_LIB=$PWD/raytracer.glsl_amd/librtgl_amd_$v.so
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/ab.json 2> gpurun_out/ab.err || (tail -5 gpurun_out/ab.err; exit 1)
python - $v <<'PY'
import json,sys; d=json.load(open('gpurun_out/ab.json')); print(sys.argv[1], '->', round(d['value'],2),'Mpaths/s', round(d['ms_per_step'],2),'ms', 'avg launch ms', round(d['roofline']['avg_launch_ms'],3), 'cand', d['counters_per_frame']['candidates'])
PY
done
export RTGL_AMD_LIB=$PWD/raytracer.glsl_amd/librtgl_amd_nop0.so
python scripts/dbg_cand.py 2>&1 | tail -4
