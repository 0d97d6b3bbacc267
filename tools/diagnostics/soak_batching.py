# N frames of a configuration rendered frame by frame and again with frame_batch = B (for each B given): the accumulated images must be equal bit for bit
# usage: python tools/diagnostics/soak_batching.py [config] [frames] [B ...]
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import raytracer_glsl_amd as rt
sc = rt.scenes
name = sys.argv[1] if len(sys.argv) > 1 else "C2"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 240
batches = [int(a) for a in sys.argv[3:]] or [4, 16]
cfg = sc.CONFIGS[name]; W, H = cfg["width"], cfg["height"]; scene = cfg["scene"](); base = cfg["params"]()
g = sc.GlibcRand(0); ps = [base.replace(frames=f, random=g.rand()) for f in range(1, n + 1)]
def run(b):
    ctx = rt.host.Context(W, H); ctx.set_option("frame_batch", b); ctx.upload_scene(scene)
    t0 = time.perf_counter()
    for p in ps: ctx.render(p, sync=False)
    ctx.synchronize(); dt = time.perf_counter() - t0
    img = ctx.read_image(); ctx.close()
    return img, dt
ref, dt = run(1)
print(f"{name}: {n} frames frame by frame in {dt * 1e3 / n:.3f} ms per frame", flush=True)
for b in batches:
    img, dt = run(b)
    print(f"{name}: frame_batch {b}: {dt * 1e3 / n:.3f} ms per frame, pixels differing from frame by frame: {int((img.view(np.uint32) != ref.view(np.uint32)).any(axis=2).sum())}", flush=True)
