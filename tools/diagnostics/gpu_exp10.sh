set -e
echo "== shipped build"; N=40 timeout -k 10 400 python tools/diagnostics/flaky_multi.py | grep "differing images"
echo "== W = 1 claims the whole register file"; RTGL_AMD_LIB=$PWD/raytracer.glsl_amd/librtgl_amd_full.so N=40 timeout -k 10 400 python tools/diagnostics/flaky_multi.py | grep "differing images"
