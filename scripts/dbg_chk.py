# checksum of every minimum the broad phase examines (MF_CHECKSUM build, debug_skip_exact = 5), same frame several times
import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
import raytracer_glsl_amd as rt
sc = rt.scenes
cfg = sc.CONFIGS["C2"]; scene = cfg["scene"]()
ctx = rt.host.Context(cfg["width"], cfg["height"])
for k, v in (("kernel", 3), ("mf_group_quads", 1), ("counters", 1), ("debug_skip_exact", 5)): ctx.set_option(k, v)
ctx.upload_scene(scene)
p = cfg["params"]().replace(frames=1, random=sc.GlibcRand(0).rand(), max_bounce=1)
for i in range(10):
    ctx.render(p); c = ctx.counters(); print("candidates", c["candidates"], flush=True)
ctx.close()
