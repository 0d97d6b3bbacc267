# reads a rocprofv3 kernel trace (csv) of tools/diagnostics/solo_frames.py and prints, per bounce, the mean duration of the scan launches
# (and per frame the totals of the other kernels).  usage: python per_bounce_trace.py <dir with *_kernel_trace.csv> <bounces per frame>
import sys, csv, glob, collections
d, nb = sys.argv[1], int(sys.argv[2])
f = max(glob.glob(d + "/**/*_kernel_trace.csv", recursive=True))
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
scan = [r for r in rows if "scan_solo_kernel" in r["Kernel_Name"]]
frames = len(scan) // nb
skip = 3
per = [[] for _ in range(nb)]
for i, r in enumerate(scan):
    fr, b = divmod(i, nb)
    if fr >= skip and fr < frames: per[b].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("frames", frames - skip, "| scan per bounce, us:", " ".join(f"{sum(p) / max(len(p), 1):.1f}" for p in per), "| sum", round(sum(sum(p) / max(len(p), 1) for p in per), 1))
other = collections.defaultdict(float)
for r in rows:
    if "scan_solo_kernel" not in r["Kernel_Name"]: other[r["Kernel_Name"].split("(")[0][-40:]] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
print("other kernels, us per frame (all frames):", {k: round(v / frames, 1) for k, v in sorted(other.items(), key=lambda kv: -kv[1])[:8]})
