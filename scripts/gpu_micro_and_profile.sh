set -e
mkdir -p gpurun_out
./tools/valu_rate | tee gpurun_out/valu_rate.txt
bash scripts/gpu_profile.sh r1_lds2 --wf-mode 1 --wf-rays 2
