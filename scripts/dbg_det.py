import sys, os, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import raytracer_glsl_amd as rt
sc = rt.scenes
cfg = sc.CONFIGS["C2"]; W, H = cfg["width"], cfg["height"]; scene = cfg["scene"]()
nb = int(os.environ.get("NB", "1")); runs = int(os.environ.get("RUNS", "8"))
def run(opts):
    ctx = rt.host.Context(W, H)
    for k, v in opts: ctx.set_option(k, v)
    ctx.set_option("counters", 1)
    ctx.upload_scene(scene)
    p = cfg["params"]().replace(frames=1, random=sc.GlibcRand(0).rand(), max_bounce=nb)
    out = []
    for i in range(runs):
        ctx.render(p.replace(reset_flag=1) if False else p)   # same frame again and again (frames=1 => previous image ignored)
        c = ctx.counters(); img = ctx.read_image()
        out.append((c["candidates"], int(img.view(np.uint32).astype(np.uint64).sum() & 0xffffffff)))
    ctx.close(); return out
print("k2", run((("kernel", 2),)))
for q in (1, 16):
    print("k3 Q", q, run((("kernel", 3), ("mf_group_quads", q))), flush=True)
