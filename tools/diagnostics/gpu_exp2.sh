set -e
RTGL_AMD_LIB=$PWD/raytracer.glsl_amd/librtgl_amd_stamps.so timeout -k 10 120 python tools/diagnostics/solo_frames.py C2 10 2>&1 | tee gpurun_out/r2_stamps_C2.txt
timeout -k 10 300 python bench.py --config C4 --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/exp_c4.json 2>/dev/null
python -c "import json; d=json.load(open('gpurun_out/exp_c4.json')); print('C4', round(d['value'],2), 'Mpaths/s', round(d['ms_per_step'],1), 'ms')"
timeout -k 10 120 python tools/diagnostics/solo_frames.py C5 30
timeout -k 10 120 python tools/diagnostics/solo_frames.py C2 100
