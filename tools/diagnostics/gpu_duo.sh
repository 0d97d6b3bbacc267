set -e
mkdir -p gpurun_out
make -s -C oracle liboracle.so
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r2_pytest_duo.log 2>&1 || (tail -40 gpurun_out/r2_pytest_duo.log; exit 1)
tail -2 gpurun_out/r2_pytest_duo.log
for w in 0 1 2; do echo "scan_waves $w"; RTGL_AMD_SCAN_WAVES=$w timeout -k 10 120 python tools/diagnostics/solo_frames.py C2 100; done
timeout -k 10 120 python tools/diagnostics/solo_frames.py C5 20
timeout -k 10 120 python tools/diagnostics/solo_frames.py C4 10
FRAMES=300 timeout -k 10 600 python tools/diagnostics/soak_determinism.py | tee gpurun_out/r2_soak_c2.txt
CFG=C5 FRAMES=12 timeout -k 10 300 python tools/diagnostics/soak_determinism.py | tee gpurun_out/r2_soak_c5.txt
