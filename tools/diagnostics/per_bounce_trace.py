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
# timeline of a frame (argument 3 = "timeline"): per position in the frame the kernel, its mean duration and the mean gap since the previous kernel's end
if len(sys.argv) > 3 and sys.argv[3] == "timeline":
    gens = [i for i, r in enumerate(rows) if "generate_rays_kernel" in r["Kernel_Name"]]
    spans = [rows[a:b] for a, b in zip(gens[skip:-1], gens[skip + 1:])]
    n = min(len(s) for s in spans); spans = [s for s in spans if len(s) == n]
    tot_d = tot_g = 0.0
    for k in range(n):
        dur = sum(int(s[k]["End_Timestamp"]) - int(s[k]["Start_Timestamp"]) for s in spans) / len(spans) / 1e3
        gap = sum(int(s[k]["Start_Timestamp"]) - int(s[k - 1]["End_Timestamp"]) for s in spans) / len(spans) / 1e3 if k else 0.0
        tot_d += dur; tot_g += gap
        print(f"  {k:2d} {spans[0][k]['Kernel_Name'].split('(')[0][-36:]:36s} {dur:7.1f} us   gap before {gap:5.1f}")
    print(f"  kernels {tot_d:.1f} us + gaps {tot_g:.1f} us per frame ({len(spans)} frames)")
