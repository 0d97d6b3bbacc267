# two INDEPENDENT contexts rendering concurrently on one device (frames enqueued without waiting, then both waited for), one with the
# one-wave-per-SIMD kernel-4 scan and one with another scan: whose image goes wrong?
import sys, os, numpy as np
os.environ.setdefault("RTGL_AMD_PRIVATE_STREAMS", "1")      # one stream per context, as before the contexts of a device shared one: this script is about concurrency
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import raytracer_glsl_amd as rt
sc = rt.scenes
W, H = 328, 204
scene = sc.scene_mesh(30, 10, env_size=32); base = sc.params_c2()
g = sc.GlibcRand(0); plist = [base.replace(frames=f, random=g.rand()) for f in range(1, 4)]
def make(opts):
    ctx = rt.host.Context(W, H)
    for k, v in opts: ctx.set_option(k, v)
    ctx.upload_scene(scene); return ctx
def frames(ctxs):
    for c in ctxs: c.write_image(np.zeros((H, W, 4), np.float32))
    for p in plist:
        for c in ctxs: c.render(p, sync=False)
    for c in ctxs: c.synchronize()
    return [c.read_image() for c in ctxs]
r = make((("kernel", 2),)); ref = frames([r])[0]; r.close()
N = int(os.environ.get("N", "150"))
for names, optsets in ((("kernel 4, one wave", "kernel 2"), ((("scan_waves", 1),), (("kernel", 2),))),
                       (("kernel 4, one wave", "kernel 4, one wave"), ((("scan_waves", 1),), (("scan_waves", 1),))),
                       (("kernel 4, one wave", "kernel 4, two waves"), ((("scan_waves", 1),), (("scan_waves", 2),))),
                       (("kernel 4, two waves", "kernel 2"), ((("scan_waves", 2),), (("kernel", 2),))),
                       (("kernel 2", "kernel 2"), ((("kernel", 2),), (("kernel", 2),)))):
    ctxs = [make(o) for o in optsets]; bad = [0, 0]
    for it in range(N):
        imgs = frames(ctxs)
        for i, img in enumerate(imgs):
            d = (img.view(np.uint32) != ref.view(np.uint32)).any(axis=2)
            if d.any():
                bad[i] += 1
                if bad[i] <= 2: ys, xs = np.nonzero(d); print("   ", names[i], "(beside", names[1 - i] + ") iteration", it, ":", int(d.sum()), "pixels from", (int(ys[0]), int(xs[0])), flush=True)
    for c in ctxs: c.close()
    print(names[0], "beside", names[1], ": wrong images", bad[0], "and", bad[1], "of", N, flush=True)
