# a small mesh scene rendered with kernel 2 (fp32 scan) and with kernel 4 under every combination of waves per SIMD / culling / chunk size:
# differing pixels per combination (all must be 0)
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import raytracer_glsl_amd as rt
sc = rt.scenes
W, H = int(os.environ.get("W", "328")), int(os.environ.get("H", "204"))
scene = sc.scene_mesh(30, 10, env_size=32); base = sc.params_c2()
g = sc.GlibcRand(0); plist = [base.replace(frames=f, random=g.rand()) for f in range(1, 4)]
def run(opts, **kw):
    ctx = rt.host.Context(W, H, **kw)
    for k, v in opts: ctx.set_option(k, v)
    ctx.set_option("counters", 1); ctx.upload_scene(scene)
    for p in plist: ctx.render(p)
    img, cnt = ctx.read_image(), ctx.counters(); ctx.close(); return img, cnt
ref, cref = run((("kernel", 2),))
for waves in (1, 2):
    for cull in (0, 1, 2):
        for chunk in (32, 3):
            img, cnt = run((("kernel", 4), ("scan_waves", waves), ("cull", cull), ("mf_chunk_quads", chunk)))
            print("waves", waves, "cull", cull, "chunk", chunk, "differing pixels", int((img.view(np.uint32) != ref.view(np.uint32)).any(axis=2).sum()), "candidates", cnt["candidates"], "culled", cnt["culled_tests"], flush=True)
img, cnt = run((("kernel", 4),), devices=[0, 0], strip_rows=8)
print("two parts on one device: differing pixels", int((img.view(np.uint32) != ref.view(np.uint32)).any(axis=2).sum()))
