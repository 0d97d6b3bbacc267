# renders N frames of a configuration and closes the context (a -DRT_SOLO_STAMPS build prints its per-bounce cycle breakdown at
# rtgl_destroy).  usage: [RTGL_AMD_LIB=...] python tools/diagnostics/solo_frames.py [C2] [frames]
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import raytracer_glsl_amd as rt
sc = rt.scenes
name = sys.argv[1] if len(sys.argv) > 1 else "C2"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
cfg = sc.CONFIGS[name]; W, H = cfg["width"], cfg["height"]; scene = cfg["scene"](); base = cfg["params"]()
ctx = rt.host.Context(W, H, device=0)
for kv in filter(None, os.environ.get("OPTS", "").split(",")):      # OPTS=cull=1,scan_waves=1 ...
    k, v = kv.split("="); ctx.set_option(k, int(v))
ctx.upload_scene(scene)
if os.environ.get("KERNEL_TIMING"): ctx.set_option("kernel_timing", 1)      # HIP events around every scan launch: total per frame printed below
g = sc.GlibcRand(0); ps = [base.replace(frames=f, random=g.rand()) for f in range(1, n + 4)]
for p in ps[:3]: ctx.render(p, sync=False)
ctx.synchronize(); t0 = time.perf_counter()
for p in ps[3:]: ctx.render(p, sync=False)
ctx.synchronize(); dt = (time.perf_counter() - t0) / n
print(f"{name}: {dt * 1e3:.3f} ms per frame = {(W // 8 * 8) * (H // 8 * 8) / dt / 1e6:.1f} Mpaths/s", flush=True)
if os.environ.get("KERNEL_TIMING"):
    t = ctx.accumulated_timing(); print({k: (round(v / (n + 3), 4) if isinstance(v, float) else v) for k, v in t.items()}, "(ms per frame over", n + 3, "frames)", flush=True)
ctx.close()
