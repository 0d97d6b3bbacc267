# frames of ONE rank's strips (world N, rank R) -- run under rocprofv3 --kernel-trace --stats to see where a rank's frame goes
# usage: python tools/diagnostics/rank_frames.py [world] [rank] [frames]
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import raytracer_glsl_amd as rt
sc = rt.scenes
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rank = int(sys.argv[2]) if len(sys.argv) > 2 else 0
n = int(sys.argv[3]) if len(sys.argv) > 3 else 50
cfg = sc.CONFIGS[os.environ.get("CONFIG", "C2")]; W, H = cfg["width"], cfg["height"]; scene = cfg["scene"](); base = cfg["params"]()
ctx = rt.host.Context(W, H, device=0, rank=rank, world=world, strip_rows=8)
ctx.upload_scene(scene)
g = sc.GlibcRand(0); ps = [base.replace(frames=f, random=g.rand()) for f in range(1, n + 6)]
for p in ps[:5]: ctx.render(p, sync=False)
ctx.synchronize(); t0 = time.perf_counter()
for p in ps[5:]: ctx.render(p, sync=False)
ctx.synchronize(); dt = (time.perf_counter() - t0) / n
print(f"world {world} rank {rank}: {dt * 1e3:.3f} ms per frame, context {ctx.get_option('device_mbytes')} MiB, candidate region {ctx.get_option('cand_region_pairs')} pairs", flush=True)
ctx.close()
