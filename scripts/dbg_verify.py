import sys, os, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import raytracer_glsl_amd as rt
sc = rt.scenes
cfg = sc.CONFIGS["C2"]; W, H = cfg["width"], cfg["height"]; scene = cfg["scene"]()
for q in (1, 16):
    ctx = rt.host.Context(W, H)
    for k, v in (("kernel", 3), ("mf_group_quads", q), ("counters", 1), ("debug_skip_exact", 3)): ctx.set_option(k, v)
    ctx.upload_scene(scene)
    g = sc.GlibcRand(0)
    for f in range(1, 3):
        ctx.render(cfg["params"]().replace(frames=f, random=g.rand()))
    print("Q", q, flush=True); sys.stderr.flush()
    cnt = ctx.counters(); ctx.close()
