# C4: group-size sweep under the two-wave scan, one against two waves per SIMD, cycle stamps
set -e
mkdir -p gpurun_out
for g in 8 16 32; do
  timeout -k 10 300 python bench.py --config C4 --steps 10 --warmup 3 --no-cpu-baseline --mf-group-quads $g > gpurun_out/exp5_c4_g$g.json 2>/dev/null
  python -c "import json; d=json.load(open('gpurun_out/exp5_c4_g$g.json')); print('C4 group', $g, round(d['value'],2), 'Mpaths/s candidates', d['counters_per_frame']['candidates'])"
done
RTGL_AMD_SCAN_WAVES=1 timeout -k 10 120 python tools/diagnostics/solo_frames.py C4 8
RTGL_AMD_SCAN_WAVES=2 timeout -k 10 120 python tools/diagnostics/solo_frames.py C4 8
RTGL_AMD_LIB=$PWD/raytracer.glsl_amd/librtgl_amd_stamps.so timeout -k 10 120 python tools/diagnostics/solo_frames.py C4 5 2>&1 | tee gpurun_out/r2_stamps_C4.txt
RTGL_AMD_LIB=$PWD/raytracer.glsl_amd/librtgl_amd_stamps.so timeout -k 10 120 python tools/diagnostics/solo_frames.py C2 10 2>&1 | tee gpurun_out/r2_stamps_C2_auto.txt
