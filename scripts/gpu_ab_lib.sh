# usage: bash scripts/gpu_ab_lib.sh <suffix> ...   -- bench the default library and librtgl_amd_<suffix>.so side by side
set -e
mkdir -p gpurun_out
run() {
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/ab.json 2> gpurun_out/ab.err || (tail -5 gpurun_out/ab.err; exit 1)
python - "$1" <<'PY'
import json,sys; d=json.load(open('gpurun_out/ab.json')); print(sys.argv[1], '->', round(d['value'],2),'Mpaths/s', round(d['ms_per_step'],2),'ms', 'avg launch ms', round(d['roofline']['avg_launch_ms'],3), 'cand', d['counters_per_frame']['candidates'])
PY
}
run default
for v in "$@"; do RTGL_AMD_LIB=$PWD/raytracer.glsl_amd/librtgl_amd_$v.so run $v; done
run default
