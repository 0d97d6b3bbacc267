# determinism soak for the default scan: N frames of a configuration rendered twice with kernel 4 (and once with one-quad groups
# and the cull on every bounce) and once with kernel 2 (fp32 scan);
# per-frame survivor counts must repeat exactly and the accumulated images must be bit-identical
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import raytracer_glsl_amd as rt
sc = rt.scenes
cfg = sc.CONFIGS[os.environ.get("CFG", "C2")]; W, H = cfg["width"], cfg["height"]; scene = cfg["scene"]()
N = int(os.environ.get("FRAMES", "24"))
def run(opts):
    ctx = rt.host.Context(W, H)
    for k, v in opts: ctx.set_option(k, v)
    ctx.set_option("counters", 1); ctx.upload_scene(scene)
    g = sc.GlibcRand(0); cands = []
    for f in range(1, N + 1):
        ctx.render(cfg["params"]().replace(frames=f, random=g.rand())); cands.append(ctx.counters()["candidates"])
    img = ctx.read_image(); ctx.close(); return cands, img
K = int(os.environ.get("KERNEL", "4"))
c1, i1 = run((("kernel", K),)); c2, i2 = run((("kernel", K),)); c3, i3 = run((("kernel", K), ("mf_group_quads", 1), ("cull", 2))); _, ref = run((("kernel", 2),))
print("frames", N, "survivor counts repeat:", c1 == c2, "sum", sum(c1))
print("kernel", K, "run 1 vs run 2 differing pixels:", int((i1.view(np.uint32) != i2.view(np.uint32)).any(axis=2).sum()))
print("kernel", K, "vs kernel 2 differing pixels:", int((i1.view(np.uint32) != ref.view(np.uint32)).any(axis=2).sum()), "(Q=1:", int((i3.view(np.uint32) != ref.view(np.uint32)).any(axis=2).sum()), ")")
