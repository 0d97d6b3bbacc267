# one environment variable over a list of values on some configurations: VAR=name VALS="a b c" [CFGS="C2 C5 C4"] [FRAMES=40]
set -e
mkdir -p gpurun_out
: > gpurun_out/sweep.txt
for cfg in ${CFGS:-C2 C5 C4}; do
  for v in $VALS; do
    echo "== $cfg $VAR=$v" >> gpurun_out/sweep.txt
    env $VAR=$v timeout -k 10 200 python tools/diagnostics/solo_frames.py $cfg ${FRAMES:-40} >> gpurun_out/sweep.txt 2>&1
  done
done
cat gpurun_out/sweep.txt
