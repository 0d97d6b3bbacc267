#!/usr/bin/env python3
"""Headline benchmark: Mpaths/s on the C2 workload of BASELINE.json (1920x1080, 8 bounces,
10,000-triangle mesh + spheres + cube map), progressive frames on N MI355X of one node.

    python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run)

A "step" is one frame: one pass of the per-pixel path-trace hot path over the whole image (1 sample
per pixel) including the running-mean accumulation and, for N>1, the frame-end gather of the tile
buffers to rank 0 over RCCL.  Scene, cube map and the accumulation image are resident in HBM before
the timed region starts.  One JSON line is printed by rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
FP32_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: peak FP32 vector
BF16_DENSE_PEAK_TFLOPS = 2500.0   # MI355X_MICROARCH.md: dense bf16 matrix peak (the headline 5 PF figure includes 2:1 sparsity)


def scene_bytes(scene) -> int:
    """S_scene of SURVEY.md 8(d3): reference-layout bytes one pass over the scene reads."""
    return (48 * scene.n_triangles + 32 * scene.spheres.shape[0] + 48 * scene.nodes.shape[0]
            + 32 * scene.materials.shape[0])


def algorithmic_bytes(width, height, counters, scene, reset=False) -> dict:
    """SURVEY.md 8(d3), per frame: image read+write (32 B/px, 16 on a reset frame) + one scene stream
    per 256-ray group per bounce pass (segments/256 groups with perfect compaction) + 16 B per
    environment lookup (four RGBA8 texels).  The scene-stream term belongs to the ray x triangle
    kernel (the dominant one); image and environment terms to the shade kernel."""
    px = (width // 8 * 8) * (height // 8 * 8)
    scan = (counters["segments"] / 256.0) * scene_bytes(scene)
    return {"scan": scan, "total": px * (16 if reset else 32) + scan + 16.0 * counters["env_lookups"]}


def cpu_baseline(rt, scene, params, width, height, budget_rows=256):
    """The oracle (C restatement, oracle/pathtrace_oracle.c) timed on this host's cores over a
    bounded, uniformly strided sample of the same frame: strips of 8 rows every `stride` rows."""
    from oracle.oracle import CpuOracle
    import numpy as np
    orc = CpuOracle()
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = min(cores, 16)      # a 1-GPU box's share of the host (the box itself may show hundreds of CPUs)
    dh = height // 8 * 8
    n_strips = max(1, budget_rows // 8)
    stride = max(8, (dh // n_strips) // 8 * 8)
    img = np.zeros((height, width, 4), np.float32)
    pixels = 0
    t0 = time.perf_counter()
    for y0 in range(0, dh, stride):
        orc.render(scene, params, img, rect=(0, y0, width // 8 * 8, min(y0 + 8, dh)), threads=cores)
        pixels += (width // 8 * 8) * (min(y0 + 8, dh) - y0)
    dt = time.perf_counter() - t0
    return {
        "value": pixels * params.samples / dt / 1e6, "unit": "Mpaths/s", "cores": cores, "kind": "port",
        "sample": f"one frame, 8-row strips every {stride} rows = {pixels} of {(width // 8 * 8) * dh} pixels, {dt:.1f} s",
    }


def relaunch_command(argv, n_gpus, port=None):
    """`python bench.py --gpus N ...` typed without a launcher: the command that starts the N ranks (one per GPU, RCCL), built before
    anything touches the GPU.  The parent only relays the child's output and return code."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port if port is not None else 29400 + os.getpid() % 2000), os.path.abspath(__file__)]
    return cmd + list(argv)


def relaunch(argv, n_gpus):
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.run(relaunch_command(argv, n_gpus), env=env)
    return proc.returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=800)      # 2.7 s of timed region on C2: clocks settle, the driver's SMI sampler sees the run
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="C2", help="C2 (headline), C4, C5 or C1")
    ap.add_argument("--kernel", type=int, default=None, help="kernel variant override (rtgl_set_option kernel)")
    ap.add_argument("--wf-rays", type=int, default=None, help="rays per lane of the wavefront kernel (1, 2, 4)")
    ap.add_argument("--wf-mode", type=int, default=None, help="triangle operand path: 0 scalar loads, 1 LDS tiles")
    ap.add_argument("--wf-chunk", type=int, default=None, help="triangles per work item of the split intersect kernel")
    ap.add_argument("--wf-early", type=int, default=None, help="leading bounces with the wave-level edge short circuit")
    ap.add_argument("--wf-packed", type=int, default=None, help="v_pk_fma_f32 ray pairs (1) or plain v_fma_f32 (0)")
    ap.add_argument("--debug-skip-exact", type=int, default=None, help="diagnostic (wrong image): 1 drops the broad-phase survivors, 2 lets nothing survive")
    ap.add_argument("--mf-chunk-quads", type=int, default=None, help="kernel 4: 40-triangle quads per LDS-resident chunk (1..32)")
    ap.add_argument("--mf-group-quads", type=int, default=None, help="kernel 4: quads sharing one local origin (1, 2, 4 .. 64)")
    ap.add_argument("--cull", type=int, default=None, help="kernel 4 packet culling: 0 off, 1 camera-ray bounce, 2 every bounce (queues as they come), 3 camera rays + binned queues (default)")
    ap.add_argument("--sort-min-rays", type=int, default=None, help="kernel 4, cull 3: a bounce's queue is binned when at least this many rays are expected")
    ap.add_argument("--debug-bounces", type=int, default=None, help="diagnostic: override the bounce limit of the configuration (not the named workload)")
    ap.add_argument("--strip-rows", type=int, default=8, help="rows per interleaved strip (multiple of 8); 8 balances the ranks to +-3%% at N = 8, 16 to +-6%%")
    ap.add_argument("--timing-period", type=int, default=None, help="frames between kernel-timed frames (default: 4)")
    ap.add_argument("--frame-batch", type=int, default=1, help="opt-in (rtgl option frame_batch): trace this many consecutive frames in one set of launches; the image "
                    "(and the gather at N > 1) then follows every batch instead of every frame, bit-identical; disables the per-launch kernel timing")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--batched-extra", choices=("auto", "on", "off"), default="auto", help="the second, separately reported region (frame_batch = 8, or 2 x N at N > 4, frames per set "
                    "of launches; reported under `frame_batched`, never as `value`): auto = on; the profiling scripts pass `off` so that a profile holds the launches of the `value` region alone")
    ap.add_argument("--sync-each-frame", action="store_true", help="diagnostic: host waits for every frame")
    ap.add_argument("--no-kernel-timing", action="store_true", help="diagnostic: no HIP events around the scan launches")
    ap.add_argument("--cpu-rows", type=int, default=1080, help="rows of the frame the CPU baseline renders (8-row strips, uniformly strided): 1080 = the whole C2 frame, ~10 s on 16 threads")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:      # typed without a launcher: start the ranks as a child job (nothing has touched the GPU yet)
        raise SystemExit(relaunch(sys.argv[1:], args.gpus))

    import torch
    import raytracer_glsl_amd as rt
    sc = rt.scenes

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE {world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    # TEST ONLY (tests/test_gpu_bench_launcher.py, a one-GPU box): RTGL_BENCH_SHARED_DEVICE=1 maps every rank to device 0.  RCCL refuses two
    # ranks on one device, so that mapping exchanges the tile buffers through the host with gloo; everything else -- the launcher, the rank
    # plumbing, the strips, the barriers, the JSON line -- is the code the driver's multi-GPU run takes.
    shared_device = os.environ.get("RTGL_BENCH_SHARED_DEVICE") == "1"
    dev_index = 0 if shared_device else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    backend = "none"
    if world > 1:
        import torch.distributed as dist
        backend = "gloo" if shared_device else "nccl"
        if shared_device:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)   # nccl == RCCL on ROCm

    cfg = sc.CONFIGS[args.config]
    W, H = cfg["width"], cfg["height"]
    scene = cfg["scene"]()
    base = cfg["params"]()
    if args.debug_bounces is not None:
        base.max_bounce = args.debug_bounces

    ctx = rt.host.Context(W, H, device=dev_index, rank=rank, world=world, strip_rows=args.strip_rows)
    ctx.upload_scene(scene)
    for key, val in (("kernel", args.kernel), ("wf_rays", args.wf_rays), ("wf_mode", args.wf_mode), ("wf_chunk", args.wf_chunk), ("wf_early", args.wf_early), ("wf_packed", args.wf_packed),
                     ("mf_chunk_quads", args.mf_chunk_quads), ("cull", args.cull), ("sort_min_rays", args.sort_min_rays), ("mf_group_quads", args.mf_group_quads), ("debug_skip_exact", args.debug_skip_exact)):
        if val is not None:
            ctx.set_option(key, val)
    gat = rt.tiling.FrameGatherer(W, H, rank, world, torch.device("cpu") if shared_device else dev, args.strip_rows)
    if not shared_device:
        ctx.bind_device_image(gat.local.data_ptr())      # render straight into the buffer the gather sends
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    # HIP events around every launch of the dominant kernel, in every 4th frame: each event pair is ~3 us of launch gap, 16 pairs per frame
    # are 1.4 % of a C2 frame at N = 1 and 7 % of a rank's frame at N = 8
    if args.frame_batch > 1:
        args.no_kernel_timing = True                      # (a context that times its launches renders frame by frame)
        ctx.set_option("frame_batch", args.frame_batch)
        assert args.steps % args.frame_batch == 0, "--steps must be a multiple of --frame-batch"
    timing_period = 0 if args.no_kernel_timing else (args.timing_period or 4)
    ctx.set_option("kernel_timing", timing_period)

    rnd = sc.GlibcRand(0)
    frame_no = [0]

    def next_params():
        frame_no[0] += 1
        return base.replace(frames=frame_no[0], random=rnd.rand())

    submitted, batch_now = [0], [args.frame_batch]

    def step(p):
        ctx.render(p, sync=False)
        submitted[0] += 1
        if submitted[0] % batch_now[0] == 0:   # (every frame unless --frame-batch: then when the batch has been submitted)
            if shared_device:              # (test mapping: tile buffer through the host, blocking gloo gather)
                gat.local[: ctx.local_rows] = torch.from_numpy(ctx.read_image())
                gat.gather()
            else:
                gat.gather(overlap=world > 1)  # N > 1: snapshot + asynchronous RCCL gather, overlapped with the next frame's render

    def barrier():
        ctx.synchronize()                  # (submits what a batching context still holds back)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(next_params())
    ctx.synchronize()
    submitted[0] = 0
    gat.finish()
    timed = [next_params() for _ in range(args.steps)]
    barrier()
    ctx.timing_reset()
    t0 = time.perf_counter()
    for p in timed:                                       # frames are queued back to back: no host sync inside the region
        step(p)
        if args.sync_each_frame:
            ctx.synchronize()
    gat.finish()                           # the last frame's exchange and un-permute belong to the timed region
    barrier()
    dt = time.perf_counter() - t0
    timed_frames = args.steps                              # frames the event timing covers
    if args.no_kernel_timing:
        frame_ms, scan_ms, scan_launches = dt * 1e3, dt * 1e3, args.steps * base.max_bounce
    else:
        t = ctx.accumulated_timing()                      # HIP events recorded on the launch stream around every scan launch
        frame_ms, scan_ms, scan_launches = t["frame_ms"], t["intersect_ms"], t["intersect_launches"]
        timed_frames = t["frames"]
        assert timed_frames == (args.steps + timing_period - 1) // timing_period
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device="cpu" if shared_device else dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # A second region, reported beside `value`, never as it: the same workload with 2 x N (at most 16) frames traced per set of launches (option
    # "frame_batch", DESIGN.md 7: bit-identical image; the image -- and at N > 1 the gather -- follows every batch instead of every frame).
    batched = None
    if args.frame_batch == 1 and args.batched_extra in ("on", "auto") and args.steps >= 8 and not args.sync_each_frame and not shared_device:
        try:                                               # (whatever happens here must not cost the line its `value`)
            B = min(max(8, 2 * world), 16, args.steps // 8 * 8 if args.steps < 16 else 16)   # eight frames per set of launches (a rank of N > 4: what two whole frames are to a single GPU)
            B = max(B, 2)
            kb = args.steps // B * B
            ctx.set_option("kernel_timing", 0)
            ctx.set_option("frame_batch", B)
            batch_now[0], submitted[0] = B, 0
            for _ in range(2 * B):
                step(next_params())
            ctx.synchronize()
            gat.finish()
            extra = [next_params() for _ in range(kb)]
            barrier()
            tb = time.perf_counter()
            for p in extra:
                step(p)
            gat.finish()
            barrier()
            dtb = time.perf_counter() - tb
            if world > 1:
                tt = torch.tensor([dtb], dtype=torch.float64, device=dev)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                dtb = float(tt.item())
            ctx.set_option("frame_batch", 1)
            ctx.set_option("kernel_timing", timing_period)
            batch_now[0] = 1
            batched = {"frame_batch": B, "steps": kb, "ms_per_step": dtb / kb * 1e3, "unit": "Mpaths/s",
                       "value": (W // 8 * 8) * (H // 8 * 8) * base.samples * kb / dtb / 1e6,
                       "note": f"OPT-IN, not the default and not `value`: the same frames, {B} traced per set of launches (rtgl option frame_batch, bit-identical image); the image "
                               "and the gather follow every batch, not every frame (a display every B-th frame); measured after the region `value` comes from, same barriers.  "
                               "More rays per launch fill the small late-bounce kernels and the direction / origin bins of the culling"}
        except Exception as e:                              # noqa: BLE001
            batched = {"error": f"{type(e).__name__}: {e}"}
            ctx.set_option("frame_batch", 1)
            batch_now[0] = 1

    # untimed: work counters of one representative frame (atomics are off in the timed region)
    ctx.set_option("counters", 1)
    ctx.render(timed[0].replace(frames=frame_no[0] + 1), sync=True)
    cnt = ctx.counters()
    ctx.set_option("counters", 0)
    if world > 1:
        ct = torch.tensor([cnt["segments"], cnt["triangle_tests"], cnt["env_lookups"], cnt["paths"], cnt["candidates"], cnt["culled_tests"]],
                          dtype=torch.float64, device="cpu" if shared_device else dev)
        dist.all_reduce(ct, op=dist.ReduceOp.SUM)
        cnt = dict(segments=int(ct[0]), triangle_tests=int(ct[1]), env_lookups=int(ct[2]), paths=int(ct[3]), candidates=int(ct[4]), culled_tests=int(ct[5]))

    if rank == 0:
        px = (W // 8 * 8) * (H // 8 * 8)
        paths = px * base.samples * args.steps
        # dominant kernel = the ray x triangle scan (rank 0's launches; its share of the frame is 1/world)
        share = 1.0 / world
        alg = algorithmic_bytes(W, H, cnt, scene)
        launches_per_frame = max(scan_launches // max(timed_frames, 1), 1)
        avg_launch_s = max(scan_ms, 1e-9) / 1e3 / max(scan_launches, 1)
        bytes_per_launch = alg["scan"] * share / launches_per_frame
        if scan_launches == 0:            # a scene without triangles has no scan launches: describe the whole frame instead
            launches_per_frame, avg_launch_s, bytes_per_launch = 1, frame_ms / 1e3 / max(timed_frames, 1), alg["total"] * share
        flops_per_launch = cnt["triangle_tests"] * 36.0 * share / launches_per_frame   # 18 fma per edge-function triple
        kname = {0: "pathtrace_mega_kernel", 1: "bounce_kernel", 2: "intersect_kernel", 4: "scan_solo_kernel"}[ctx.get_option("kernel_in_use")]
        if scan_launches == 0:
            kname = "whole frame (generate_rays + shade)"
        k = ctx.get_option("kernel_in_use")
        gtests = cnt["triangle_tests"] * share / launches_per_frame / avg_launch_s / 1e9
        tflops = flops_per_launch / avg_launch_s / 1e12
        hbm = {"achieved": bytes_per_launch / avg_launch_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
               "frac": bytes_per_launch / avg_launch_s / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": bytes_per_launch,
               "note": "SURVEY 8(d3) accounting: one scene stream per 256 live rays per bounce; the scene lives in L2/LDS, so this fraction is small by construction"}
        traffic, traffic_src = None, None
        tf = os.path.join(ROOT, "profiles", "hbm_traffic.json")     # written by tools/diagnostics/gpu_profile.sh (rocprofv3 PMC passes), keyed "<config>/<kernel>"
        if os.path.exists(tf) and world == 1:
            ent = json.load(open(tf)).get(f"{args.config}/{kname}")
            if ent:
                traffic, traffic_src = ent["hbm_bytes_per_launch"], f"committed rocprofv3 PMC profile ({ent.get('profile', 'profiles/')}), not measured in this run"
        if k == 4:
            # The dominant kernel issues one v_mfma_f32_32x32x16_bf16 (32,768 flop) per 320 ray x triangle tests (10 triangles x 32
            # rays) plus 8 VALU instructions examining its 16 results.  Matrix and vector instructions share the SIMD's issue port:
            # the bound is matrix/vector ISSUE, priced against the dense bf16 MFMA peak (one such MFMA per 32 cycles per SIMD).
            n_simd, clk = 1024.0, 2.4e9
            products_per_launch = cnt["triangle_tests"] * share / launches_per_frame / 320.0
            executed_per_launch = (cnt["triangle_tests"] - cnt["culled_tests"]) * share / launches_per_frame / 320.0   # packet culling skips the rest
            executed_tflops = executed_per_launch * 32768.0 / avg_launch_s / 1e12
            effective_tflops = products_per_launch * 32768.0 / avg_launch_s / 1e12
            # matrix-pipe occupancy and shader clock of the same command under rocprofv3 / in the stamps build: committed profile, not this run
            derived = None
            df = os.path.join(ROOT, "profiles", "pmc_derived.json")        # tools/diagnostics/summarize_profile.py
            if os.path.exists(df) and world == 1:
                derived = json.load(open(df)).get(args.config)
            roof = {"bound": "mfma", "achieved": executed_tflops, "peak": BF16_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": executed_tflops / BF16_DENSE_PEAK_TFLOPS,
                    "traffic": traffic, "traffic_source": traffic_src, "kernel": kname, "launches_per_frame": launches_per_frame, "avg_launch_ms": avg_launch_s * 1e3,
                    "definition": "achieved = matrix flops of the v_mfma_f32_32x32x16_bf16 instructions actually ISSUED per launch (32,768 flop per product of 10 triangles x 32 rays; "
                                  "tests a granule skips because all of its rays are certified rejections are not counted) / average launch duration",
                    "products_per_launch": executed_per_launch, "culled_fraction": cnt["culled_tests"] / max(cnt["triangle_tests"], 1),
                    "cycles_per_product": n_simd * clk * avg_launch_s / max(executed_per_launch, 1.0),
                    "frac_of_mfma_issue_peak": executed_per_launch / avg_launch_s / (n_simd * clk / 32.0),
                    "issue_floor_cycles_per_product": 44.0,
                    "mfma_busy": derived,
                    "effective": {"products_per_launch": products_per_launch, "tflops": effective_tflops, "frac": effective_tflops / BF16_DENSE_PEAK_TFLOPS,
                                  "cycles_per_product": n_simd * clk * avg_launch_s / max(products_per_launch, 1.0),
                                  "note": "algorithmic equivalent, NOT an achieved-of-peak figure: every ray x triangle test the reference performs priced as if it had been "
                                          "multiplied, including the tests packet culling never issues (this can exceed 1)"},
                    "note": "dominant kernel launch = packet culling (culled bounces) + scan + narrow phase, timed together.  bound = matrix/vector issue of a SIMD: 32 cycles of matrix pipe per product; 44 cycles of issue "
                            "(MFMA 8 + 9 VALU x 4) measured for the bare instruction stream of one wave, 33 with two waves per SIMD, 38-40 with the loop's scalar instructions (tools/scan_stage_rate.hip); cycles are counted at the "
                            "nominal 2.4 GHz (the clock the chip holds in this loop: mfma_busy.in_kernel_clock_ghz).  SURVEY 8(d3)'s fp32-VALU and HBM rows are superseded for this kernel (DESIGN.md 6)",
                    "hbm": hbm}
            compute = {"pipe": "bf16 MFMA broad phase + fp32 VALU examination (one issue port per SIMD)", "algorithmic_tflops": tflops, "flop_per_test": 36,
                       "gtests_per_s": gtests, "scan_share_of_frame": scan_ms / max(frame_ms, 1e-9)}
        else:
            roof = {"bound": "hbm", "traffic": traffic, "traffic_source": traffic_src, "kernel": kname, "launches_per_frame": launches_per_frame,
                    "avg_launch_ms": avg_launch_s * 1e3, **hbm}
            compute = {"pipe": "fp32 VALU", "achieved": tflops, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tflops / FP32_PEAK_TFLOPS,
                       "flop_per_test": 36, "gtests_per_s": gtests, "scan_share_of_frame": scan_ms / max(frame_ms, 1e-9)}
        out = {
            "metric": "Mpaths/s at 1920x1080, 8 bounces, 10k tris" if args.config == "C2" else f"Mpaths/s ({args.config})",
            "value": paths / dt / 1e6, "unit": "Mpaths/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic", "rccl_ranks": (dist.get_world_size() if world > 1 else 1), "collective_backend": backend,
            "config": {"workload": f"{args.config}: {W}x{H}, {base.max_bounce} bounces, {scene.n_triangles} triangles + "
                                   f"{scene.spheres.shape[0]} spheres, cube map {scene.env.shape[1] if scene.env is not None else 0}^2, "
                                   f"1 spp/frame progressive, dof={base.use_dof}",
                       "parallelism": ("TEST MAPPING (RTGL_BENCH_SHARED_DEVICE=1): every rank on device 0, tile buffers through the host, gloo -- not a multi-GPU result; " if shared_device else "") + f"{world} GPU(s), {args.strip_rows}-row strips interleaved, gather to rank 0 every " + ("frame" if args.frame_batch == 1 else f"batch of {args.frame_batch} frames") + (" (asynchronous, overlapped with the next frame)" if world > 1 else ""),
                       "frame_batch": args.frame_batch,
                       "kernel": ctx.get_option("kernel_in_use"), "wf_rays": ctx.get_option("wf_rays"), "wf_mode": ctx.get_option("wf_mode"), "wf_chunk": ctx.get_option("wf_chunk"), "wf_early": ctx.get_option("wf_early"), "wf_packed": ctx.get_option("wf_packed"),
                       "mf_sets": ctx.get_option("mf_sets"), "mf_group_quads": ctx.get_option("mf_group_quads"), "mf_chunk_quads": ctx.get_option("mf_chunk_quads")},
            "roofline": roof,
            "compute": compute,
            "counters_per_frame": cnt,
        }
        if batched:
            out["frame_batched"] = batched
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(rt, scene, timed[0], W, H, args.cpu_rows)
            # the reference itself cannot travel to the GPU box: its llvmpipe timing is measured in the build container by
            # oracle/time_reference.py and carried here as a second, stated baseline
            rf = os.path.join(ROOT, "profiles", "llvmpipe_reference_timing.json")
            if os.path.exists(rf):
                rj = json.load(open(rf))
                pick = [c for c in rj["cases"] if c["case"].startswith("C2" if args.config != "C1" else "C1:")]
                out["cpu_baseline_reference"] = {
                    "kind": "reference", "unit": "Mpaths/s", "cores": rj["cores"], "renderer": rj["renderer"], "measured_in": rj["measured_in"],
                    "value": max(c["reference_llvmpipe_mpaths_per_s"] for c in pick) if pick else None,
                    "cases": {c["case"]: round(c["reference_llvmpipe_mpaths_per_s"], 5) for c in pick}, "note": rj["note"]}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
