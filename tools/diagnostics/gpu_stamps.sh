# cycle stamps of the scan waves (make stamps), per bounce: CFG with the default options and with OPTS2 (default cull=1)
set -e
mkdir -p gpurun_out
CFG=${CFG:-C2}
RTGL_AMD_LIB=$PWD/raytracer.glsl_amd/librtgl_amd_stamps.so timeout -k 10 200 python tools/diagnostics/solo_frames.py $CFG 20 > gpurun_out/stamps_${CFG}.txt 2>&1
RTGL_AMD_LIB=$PWD/raytracer.glsl_amd/librtgl_amd_stamps.so OPTS=${OPTS2:-cull=1} timeout -k 10 200 python tools/diagnostics/solo_frames.py $CFG 20 > gpurun_out/stamps_${CFG}_alt.txt 2>&1
