# renders rank 0's strips of an 8-rank C2 job for a few frames (no exchange): run under rocprofv3 --kernel-trace to see
# the per-launch durations and gaps a rank has at N = 8
import sys, os
sys.path.insert(0, os.getcwd())
import raytracer_glsl_amd as rt
sc = rt.scenes
cfg = sc.CONFIGS["C2"]; scene = cfg["scene"](); base = cfg["params"]()
ctx = rt.host.Context(cfg["width"], cfg["height"], device=0, rank=0, world=int(os.environ.get("WORLD", "8")), strip_rows=8)
ctx.upload_scene(scene)
g = sc.GlibcRand(0)
for f in range(1, 9):
    ctx.render(base.replace(frames=f, random=g.rand()), sync=False)
ctx.synchronize(); ctx.close()
