# three INDEPENDENT tiled contexts (rank 0..2 of 3, 8-row strips) rendering concurrently on one device, no gather: does a rank's own
# image go wrong?  (tools/diagnostics/flaky_multi.py is the same through rtgl_create_multi and its gather)
import sys, os, numpy as np
os.environ.setdefault("RTGL_AMD_PRIVATE_STREAMS", "1")      # one stream per context, as before the contexts of a device shared one: this script is about concurrency
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import raytracer_glsl_amd as rt
sc = rt.scenes
W, H = 328, 204
scene = sc.scene_mesh(30, 10, env_size=32); base = sc.params_c2()
g = sc.GlibcRand(0); plist = [base.replace(frames=f, random=g.rand()) for f in range(1, 4)]
r = rt.host.Context(W, H); r.set_option("kernel", 2); r.upload_scene(scene)
for p in plist: r.render(p)
ref = r.read_image(); r.close()
WORLD = int(os.environ.get("WORLD", "3")); N = int(os.environ.get("N", "100")); COUNTERS = int(os.environ.get("COUNTERS", "1")); CONCURRENT = int(os.environ.get("CONCURRENT", "1"))
for name, opts in (("kernel 4, one wave", (("scan_waves", 1),)), ("kernel 4, two waves", (("scan_waves", 2),)), ("kernel 2", (("kernel", 2),))):
    bad = 0; seen = {}
    for it in range(N):
        ctxs = [rt.host.Context(W, H, rank=i, world=WORLD, strip_rows=8) for i in range(WORLD)]
        for c in ctxs:
            for k, v in opts: c.set_option(k, v)
            if COUNTERS: c.set_option("counters", 1)
            c.upload_scene(scene)
        for p in plist:
            for c in ctxs: c.render(p, sync=not CONCURRENT)
        for c in ctxs: c.synchronize()
        cand = sum(c.counters()["candidates"] for c in ctxs) if COUNTERS else -1
        wrong_here = False
        for i, c in enumerate(ctxs):
            img = c.read_image(); rows = c.global_rows()
            d = (img.view(np.uint32) != ref[rows].view(np.uint32)).any(axis=2)
            if d.any():
                bad += 1; wrong_here = True
                ys, xs = np.nonzero(d)
                if bad <= 6: print("   ", name, "iteration", it, "rank", i, ":", int(d.sum()), "pixels; local rows", sorted(set(int(y) for y in ys)), "x", int(xs.min()), "..", int(xs.max()), "global row of the first", int(rows[ys[0]]), flush=True)
        seen.setdefault((cand, wrong_here), 0); seen[(cand, wrong_here)] += 1
        for c in ctxs: c.close()
    print("    (survivors counted by the scan, some image wrong) -> iterations:", seen)
    print(name, ": wrong rank images", bad, "of", N * WORLD, "(world", WORLD, "counters", COUNTERS, "concurrent", CONCURRENT, ")", flush=True)
