import sys, os, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import raytracer_glsl_amd as rt
sc = rt.scenes
cfg = sc.CONFIGS["C2"]; W, H = cfg["width"], cfg["height"]; scene = cfg["scene"]()
def render(opts, frames=1, nb=None):
    ctx = rt.host.Context(W, H); 
    for k, v in opts: ctx.set_option(k, v)
    ctx.upload_scene(scene)
    g = sc.GlibcRand(0)
    for f in range(1, frames + 1):
        p = cfg["params"]().replace(frames=f, random=g.rand())
        if nb: p = p.replace(max_bounce=nb)
        ctx.render(p)
    img = ctx.read_image(); ctx.close(); return img
nb = 1
ref = render((("kernel", 2),), nb=nb)
ref2 = render((("kernel", 0),), nb=nb)
print("k2 vs k0", int((ref.view(np.uint32) != ref2.view(np.uint32)).any(axis=2).sum()))
for name, opts in [("k3 Q1 run%d" % i, (("kernel", 3), ("mf_group_quads", 1))) for i in range(12)] + [("k3 Q16 run%d" % i, (("kernel", 3), ("mf_group_quads", 16))) for i in range(12)]:
    img = render(opts, nb=nb)
    neq = (img.view(np.uint32) != ref.view(np.uint32)).any(axis=2)
    idx = np.argwhere(neq)[:4]
    print(name, "mismatch px", int(neq.sum()), idx.tolist(), flush=True)
