# compute side of strong scaling, measured on ONE GPU: the frame time of rank r's strips for world sizes 1, 2, 4, 8
# (no exchange; what is left of the scaling loss is the gather).  python scripts/dbg_scale.py
import sys, os, time, numpy as np, torch
sys.path.insert(0, os.getcwd())
import raytracer_glsl_amd as rt
sc = rt.scenes
cfg = sc.CONFIGS["C2"]; W, H = cfg["width"], cfg["height"]; scene = cfg["scene"]()
base = cfg["params"]()
t1 = None
STRIP = int(os.environ.get("STRIP", "8"))
ALL = os.environ.get("ALL_RANKS", "0") == "1"
for world in (1, 2, 4, 8):
    for rank in (range(world) if ALL and world == 8 else sorted({0, world - 1})):
        ctx = rt.host.Context(W, H, device=0, rank=rank, world=world, strip_rows=STRIP)
        ctx.upload_scene(scene)
        g = sc.GlibcRand(0); ps = [base.replace(frames=f, random=g.rand()) for f in range(1, 34)]
        for p in ps[:3]: ctx.render(p, sync=False)
        ctx.synchronize(); t0 = time.perf_counter()
        for p in ps[3:]: ctx.render(p, sync=False)
        ctx.synchronize(); dt = (time.perf_counter() - t0) / 30 * 1e3
        if world == 1: t1 = dt
        print(f"world {world} rank {rank}: {dt:.3f} ms per frame  (ideal {t1 / world:.3f}; compute-side efficiency {t1 / world / dt:.2f})", flush=True)
        ctx.close()
