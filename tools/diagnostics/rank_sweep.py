# frame time of one rank's strips (WORLD, RANK on one GPU) of C2 for a list of option sets: python tools/diagnostics/rank_sweep.py 8 0 sort_min_rays=32768 sort_min_rays=131072,frame_batch=4
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import raytracer_glsl_amd as rt
sc = rt.scenes
world, rank = int(sys.argv[1]), int(sys.argv[2])
cfg = sc.CONFIGS[os.environ.get("CFG", "C2")]; W, H = cfg["width"], cfg["height"]; scene = cfg["scene"](); base = cfg["params"]()
for opts in sys.argv[3:] or [""]:
    ctx = rt.host.Context(W, H, device=0, rank=rank, world=world, strip_rows=8)
    for kv in filter(None, opts.split(",")):
        k, v = kv.split("="); ctx.set_option(k, int(v))
    ctx.upload_scene(scene)
    g = sc.GlibcRand(0); n = 64; ps = [base.replace(frames=f, random=g.rand()) for f in range(1, n + 17)]
    for p in ps[:16]: ctx.render(p, sync=False)
    ctx.synchronize(); t0 = time.perf_counter()
    for p in ps[16:]: ctx.render(p, sync=False)
    ctx.synchronize(); dt = (time.perf_counter() - t0) / n * 1e3
    print(f"world {world} rank {rank} [{opts}]: {dt:.3f} ms per frame", flush=True)
    ctx.close()
