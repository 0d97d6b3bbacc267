# A/B of an environment switch on the three bench configurations: VAR=name A=value B=value [CFGS="C2 C5 C4"]
set -e
mkdir -p gpurun_out
: > gpurun_out/ab.txt
for cfg in ${CFGS:-C2 C5 C4}; do
  for v in "$A" "$B"; do
    echo "== $cfg $VAR=$v" >> gpurun_out/ab.txt
    env $VAR=$v timeout -k 10 200 python tools/diagnostics/solo_frames.py $cfg ${FRAMES:-40} >> gpurun_out/ab.txt 2>&1
  done
done
cat gpurun_out/ab.txt
