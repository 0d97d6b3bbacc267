set -e
R=$(pwd); OUT=$R/gpurun_out/prof_rank8; rm -rf $OUT; mkdir -p $OUT
python tools/diagnostics/rank_frames.py 1 0 50
python tools/diagnostics/rank_frames.py 8 0 50
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/tools/diagnostics/rank_frames.py 8 0 50 > $OUT/run.log 2>&1 || (tail $OUT/run.log; exit 1)
cd $R
python - <<'PY'
import csv, glob
f = max(glob.glob("gpurun_out/prof_rank8/trace/*/*_kernel_stats.csv"))
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:8]: print(r["Name"][:60], r["Calls"], "avg_us", round(float(r["AverageNs"]) / 1e3, 2), "total_ms", round(float(r["TotalDurationNs"]) / 1e6, 2))
print("sum of kernels per frame (55 frames incl. warm-up):", round(tot / 55 / 1e3, 1), "us")
t = max(glob.glob("gpurun_out/prof_rank8/trace/*/*_kernel_trace.csv"))
k = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:30]) for r in csv.DictReader(open(t))))
gaps = [k[i + 1][0] - k[i][1] for i in range(len(k) - 1)]
gaps = [g for g in gaps if g < 200000]
import statistics
print("launch gaps: n", len(gaps), "median us", statistics.median(gaps) / 1e3, "mean us", statistics.mean(gaps) / 1e3, "sum per frame us", sum(gaps) / 55 / 1e3)
PY
