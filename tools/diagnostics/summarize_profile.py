#!/usr/bin/env python3
"""Turns a gpurun_out/prof_<tag>/ directory (written by tools/diagnostics/gpu_profile.sh on the GPU box) into the
tracked summaries under profiles/<tag>/ and, for the dominant kernel, an entry "<config>/<kernel>" of profiles/hbm_traffic.json.
    python tools/diagnostics/summarize_profile.py <tag> [config]

HBM traffic follows MI355X_MICROARCH.md (HBM / rocprofv3 section): FETCH_SIZE and WRITE_SIZE are
collected in separate --pmc passes, both are in KiB, and on gfx950 FETCH_SIZE reports exactly half of
the bytes of wide (16 B/lane) coalesced streaming reads -- which is what the scan kernel issues for its
ray and triangle-tile loads -- so the read side is doubled; WRITE_SIZE is exact."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag, config = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "C2")
src, dst = f"gpurun_out/prof_{tag}", f"profiles/{tag}"
os.makedirs(dst, exist_ok=True)
shutil.copy(max(glob.glob(f"{src}/trace/runc/*_kernel_stats.csv"), key=os.path.getmtime), f"{dst}/kernel_stats.csv")
shutil.copy(f"{src}/bench_trace.json", f"{dst}/bench_under_rocprof.json")
out = {}
for sub in ("pmc_sq", "pmc_fetch", "pmc_write", "pmc_mfma"):
    cand = glob.glob(f"{src}/{sub}/runc/*_counter_collection.csv")
    if not cand:
        continue
    f = max(cand, key=os.path.getmtime)   # gpurun merges runs of the same tag: newest wins
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    dur, calls, seen = collections.defaultdict(float), collections.Counter(), set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"]); dur[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); calls[k] += 1
    out[sub] = {k: dict(calls=calls[k], total_ms=dur[k] / 1e6, **v) for k, v in agg.items() if "rt::" in k or "pathtrace" in k}
# matrix-pipe occupancy of the scan launches and the clock rocprofv3 saw: busy cycles / (kernel cycles x 1024 SIMDs); kernel cycles =
# GRBM_GUI_ACTIVE / 8 (summed over the XCDs); effective clock = those cycles / kernel wall time
derived = {}
if "pmc_mfma" in out:
    scan = {k: v for k, v in out["pmc_mfma"].items() if "scan_solo_kernel" in k}
    busy = sum(v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) for v in scan.values()); gui = sum(v.get("GRBM_GUI_ACTIVE", 0.0) for v in scan.values())
    mops = sum(v.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0.0) for v in scan.values()); ms = sum(v["total_ms"] for v in scan.values())
    if gui > 0 and ms > 0:
        derived = {"mfma_busy_frac": busy / (gui / 8.0 * 1024.0), "mfma_busy_cycles": busy, "kernel_cycles_sum": gui / 8.0, "mfma_instructions_from_busy": busy / 32.0,
                   "mfma_mops_bf16_x512_flop": mops * 512.0, "gui_active_clock_ghz": gui / 8.0 / (ms * 1e6), "scan_launches": sum(v["calls"] for v in scan.values()),
                   "profile": dst, "source": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE (its own pass), scan_solo_kernel launches only; "
                                                     "committed profile, not measured in the bench run that quotes it"}
clk = os.path.join(src, "clock.txt")
if os.path.exists(clk):
    for line in open(clk):
        if line.startswith("rtgl clock:"):
            derived["in_kernel_clock_ghz"] = float(line.split()[-2])
            derived["in_kernel_clock_source"] = "-DRT_SOLO_STAMPS=3 build (make clock): sum of s_memtime / sum of s_memrealtime over every scan wave, 600 back-to-back frames, no profiler"
    shutil.copy(clk, f"{dst}/clock.txt")
out["derived"] = derived
json.dump(out, open(f"{dst}/pmc_summary.json", "w"), indent=1)
if derived:
    dpath = "profiles/pmc_derived.json"
    dt = json.load(open(dpath)) if os.path.exists(dpath) else {}
    dt[config] = derived
    json.dump(dt, open(dpath, "w"), indent=1)
    print(json.dumps(derived, indent=1))
# dominant kernel (largest total time in the kernel-trace stats)
stats = list(csv.DictReader(open(f"{dst}/kernel_stats.csv")))
dom = max(stats, key=lambda r: float(r["TotalDurationNs"]))["Name"].split("(")[0]
short = "scan_solo_kernel" if "scan_solo_kernel" in dom else "intersect_kernel" if "intersect_kernel" in dom else "bounce_kernel" if "bounce_kernel" in dom else "pathtrace_mega_kernel"
fetch = sum(v["FETCH_SIZE"] for k, v in out["pmc_fetch"].items() if short in k)
fcalls = sum(v["calls"] for k, v in out["pmc_fetch"].items() if short in k)
write = sum(v["WRITE_SIZE"] for k, v in out["pmc_write"].items() if short in k)
wcalls = sum(v["calls"] for k, v in out["pmc_write"].items() if short in k)
traffic = fetch * 1024 * 2 / max(fcalls, 1) + write * 1024 / max(wcalls, 1)
path = "profiles/hbm_traffic.json"
table = json.load(open(path)) if os.path.exists(path) else {}
if "kernel" in table:          # round 1 wrote one unkeyed entry
    table = {}
table[f"{config}/{short}"] = {"profile": dst, "hbm_bytes_per_launch": traffic, "launches": fcalls,
                              "fetch_kib_per_launch_raw": fetch / max(fcalls, 1), "write_kib_per_launch": write / max(wcalls, 1),
                              "correction": "FETCH_SIZE x2 (gfx950, wide coalesced reads), WRITE_SIZE exact; separate PMC passes"}
json.dump(table, open(path, "w"), indent=1)
print(open("profiles/hbm_traffic.json").read())
for r in stats[:6]:
    print(r["Name"][:70], r["Calls"], "avg_us", float(r["AverageNs"]) / 1e3)
