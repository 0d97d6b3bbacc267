# ONE run each of the three-pipeline arrangement (private streams) and of the same contexts on the shared stream,
# with AMD_LOG_LEVEL=4, to read the acquire / release fence scopes ROCclr puts into the AQL dispatch headers.  Not a stress loop: N = 1.
set -e
R=$(pwd); mkdir -p gpurun_out
for mode in private shared; do
  if [ $mode = private ]; then export RTGL_AMD_PRIVATE_STREAMS=1; else export RTGL_AMD_PRIVATE_STREAMS=0; fi
  AMD_LOG_LEVEL=4 N=1 COUNTERS=0 timeout -k 10 200 python3 tools/diagnostics/flaky_tiled.py > gpurun_out/fence_$mode.out 2> gpurun_out/fence_$mode.log || true
  echo "== $mode streams: $(grep -c 'Dispatch Header' gpurun_out/fence_$mode.log) dispatch packets logged"
  grep -o 'Dispatch Header = 0x[0-9a-f]* (type=[0-9]*, barrier=[0-9]*, acquire=[0-9]*, release=[0-9]*)' gpurun_out/fence_$mode.log | sort | uniq -c | sort -rn | head -12
  grep -o 'Barrier[A-Za-z ]*Header = 0x[0-9a-f]* ([^)]*)' gpurun_out/fence_$mode.log | sort | uniq -c | sort -rn | head -6
  grep -i -o 'HWq=0x[0-9a-f]*' gpurun_out/fence_$mode.log | sort | uniq -c | head -8
  tail -4 gpurun_out/fence_$mode.out
  grep -m3 'Dispatch Header' gpurun_out/fence_$mode.log | cut -c1-400
  rm -f gpurun_out/fence_$mode.log
done
echo "== uncached allocation probe"; ./tools/uncached_alloc_probe
