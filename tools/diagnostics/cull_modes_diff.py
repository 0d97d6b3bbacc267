# Diagnostics: one configuration rendered under several option sets; pixels differing from the first set (must be 0 everywhere).
# usage: python tools/diagnostics/cull_modes_diff.py [C2] [frames]
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import raytracer_glsl_amd as rt
sc = rt.scenes
name = sys.argv[1] if len(sys.argv) > 1 else "C2"; frames = int(sys.argv[2]) if len(sys.argv) > 2 else 2
cfg = sc.CONFIGS[name]; W, H = cfg["width"], cfg["height"]; scene = cfg["scene"](); base = cfg["params"]()
def run(opts):
    ctx = rt.host.Context(W, H, device=0)
    for k, v in opts: ctx.set_option(k, v)
    ctx.upload_scene(scene)
    g = sc.GlibcRand(0)
    for f in range(1, frames + 1): ctx.render(base.replace(frames=f, random=g.rand()), sync=False)
    img = ctx.read_image(); ctx.close(); return img
sets = [(("cull", 0),), (("cull", 1),), (("cull", 2),), (("cull", 2), ("scan_waves", 1)), (("cull", 3), ("sort_min_rays", 0)), (("cull", 3), ("sort_min_rays", 0)), (("cull", 3),),
        (("cull", 1), ("frame_batch", 2)), (("cull", 2), ("frame_batch", 2)), (("cull", 3), ("frame_batch", 2)), (("cull", 3), ("frame_batch", 2), ("scan_dynamic", 2))]
ref = None
for o in sets:
    img = run(o)
    if ref is None: ref = img; print(o, "reference"); continue
    d = (img.view(np.uint32) != ref.view(np.uint32)).any(axis=2)
    ys, xs = np.nonzero(d)
    print(o, "differing pixels:", int(d.sum()), "" if not d.any() else f"rows {ys.min()}..{ys.max()} cols {xs.min()}..{xs.max()} first {list(zip(ys[:6], xs[:6]))}", flush=True)
