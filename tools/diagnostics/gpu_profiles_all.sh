# rocprofv3 kernel-trace + PMC profiles of the bench command for C2, C4, C5 (kernel 4), then the plain bench lines
set -e
bash tools/diagnostics/gpu_profile.sh r2_k4_c2 --config C2
bash tools/diagnostics/gpu_profile.sh r2_k4_c4 --config C4
bash tools/diagnostics/gpu_profile.sh r2_k4_c5 --config C5
