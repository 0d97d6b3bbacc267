# repeats the small three-parts-on-one-device render (tests/test_gpu_fullsize.py: multi-device context) under several option sets and
# reports where an image differs from the fp32-scan reference (rows / columns / values), if it ever does
import sys, os, numpy as np
os.environ.setdefault("RTGL_AMD_PRIVATE_STREAMS", "1")      # one stream per context, as before the contexts of a device shared one: this script is about concurrency
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import raytracer_glsl_amd as rt
sc = rt.scenes
W, H = 328, 204
scene = sc.scene_mesh(30, 10, env_size=32); base = sc.params_c2()
NF = int(os.environ.get("FRAMES", "3"))
g = sc.GlibcRand(0); plist = [base.replace(frames=f, random=g.rand()) for f in range(1, NF + 1)]
def run(opts, **kw):
    ctx = rt.host.Context(W, H, **kw)
    for k, v in opts: ctx.set_option(k, v)
    ctx.set_option("counters", 1); ctx.upload_scene(scene)
    for p in plist: ctx.render(p)
    img, cnt = ctx.read_image(), ctx.counters(); ctx.close(); return img, cnt
ref, cref = run((("kernel", 2),))
# the reference after every frame: is a wrong pixel a STALE one (the value an earlier frame left)?
refs = []
ctx = rt.host.Context(W, H); ctx.set_option("kernel", 2); ctx.upload_scene(scene)
for p in plist: ctx.render(p); refs.append(ctx.read_image())
ctx.close()
N = int(os.environ.get("N", "60"))
for name, opts in (("default", ()), ("cull 0", (("cull", 0),)), ("waves 2", (("scan_waves", 2),)), ("waves 1", (("scan_waves", 1),)), ("chunk 4", (("mf_chunk_quads", 4),)), ("kernel 2", (("kernel", 2),))):
    bad = 0
    for it in range(N):
        img, cnt = run(opts, devices=[0, 0, 0], strip_rows=8)
        d = (img.view(np.uint32) != ref.view(np.uint32)).any(axis=2)
        if d.any():
            bad += 1
            ys, xs = np.nonzero(d)
            stale = [int(((img.view(np.uint32) == r.view(np.uint32)).all(axis=2) & d).sum()) for r in refs[:-1]]
            if bad <= 3: print("  of the differing pixels", stale, "carry the value the reference had after frame 1, 2, ...")
            if bad <= 3: print(" ", name, "iteration", it, ":", int(d.sum()), "pixels; (y, x):", list(zip(ys.tolist(), xs.tolist()))[:20], "| first:", img[ys[0], xs[0]], "ref", ref[ys[0], xs[0]], "| candidates", cnt["candidates"], "segments", cnt["segments"], "ref segments", cref["segments"], flush=True)
    print(name, ": differing images", bad, "of", N, flush=True)
