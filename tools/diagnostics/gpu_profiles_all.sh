# rocprofv3 kernel-trace + PMC profiles of the bench command for C2, C4, C5 (kernel 4) + the in-kernel clock
set -e
bash tools/diagnostics/gpu_profile.sh r3_k4_c2 --config C2
CLOCK_FRAMES=60 bash tools/diagnostics/gpu_profile.sh r3_k4_c4 --config C4
CLOCK_FRAMES=150 bash tools/diagnostics/gpu_profile.sh r3_k4_c5 --config C5
