set -e
mkdir -p gpurun_out
run() {
lib=$1; shift
RTGL_AMD_LIB=$PWD/raytracer.glsl_amd/$lib python bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" > gpurun_out/ab.json 2> gpurun_out/ab.err || (tail -5 gpurun_out/ab.err; exit 1)
python - $lib "$@" <<'PY'
import json,sys; d=json.load(open('gpurun_out/ab.json')); print(" ".join(sys.argv[1:]), '->', round(d['value'],2),'Mpaths/s', round(d['ms_per_step'],2),'ms', 'avg launch ms', round(d['roofline']['avg_launch_ms'],3))
PY
}
for i in 1 2; do
for lib in librtgl_amd_p1_c16.so librtgl_amd_p0_c16.so librtgl_amd_p0_c8.so librtgl_amd_p1_c8.so; do
run $lib --wf-rays 4
done
done
run librtgl_amd_p0_c8.so --wf-rays 8
run librtgl_amd_p1_c8.so --wf-rays 8
run librtgl_amd_p0_c8.so --wf-rays 2
run librtgl_amd_p0_c8.so --wf-rays 4 --wf-chunk 512
