import sys, os
sys.path.insert(0, "/root/repo")
import raytracer_glsl_amd as rt
sc = rt.scenes
for name in sys.argv[1:]:
    cfg = sc.CONFIGS[name]; ctx = rt.host.Context(cfg["width"], cfg["height"]); ctx.set_option("counters", 1); ctx.upload_scene(cfg["scene"]())
    g = sc.GlibcRand(0); ctx.render(cfg["params"]().replace(frames=1, random=g.rand())); c = ctx.counters(); ctx.close()
    print(name, "candidates", c["candidates"], end=" | ")
print()
