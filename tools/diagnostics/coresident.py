# does kernel 4 stay deterministic when OTHER kernels (copies, elementwise VALU work -- what an overlapped RCCL gather or a
# caller's own stream would bring) share the CUs with it?  Renders the same frame repeatedly while a side stream keeps the GPU
# busy, and compares survivor counts and images with a quiet run.
import sys, os, numpy as np, torch, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import raytracer_glsl_amd as rt
sc = rt.scenes
cfg = sc.CONFIGS["C2"]; scene = cfg["scene"]()
K = int(os.environ.get("KERNEL", "4"))
WORLD = int(os.environ.get("WORLD", "1")); FRAMES = int(os.environ.get("FRAMES", "24"))      # WORLD > 1: rank 0 of WORLD (the launches of a multi-GPU rank)
def run(noise):
    ctx = rt.host.Context(cfg["width"], cfg["height"], rank=0, world=WORLD, strip_rows=8)
    ctx.set_option("kernel", K); ctx.set_option("counters", 1); ctx.upload_scene(scene)
    g = sc.GlibcRand(0); out = []
    stop = [False]
    def worker():
        torch.cuda.set_device(0)
        s = torch.cuda.Stream()
        x = torch.ones(16 << 20, device="cuda"); y = torch.empty_like(x)
        with torch.cuda.stream(s):
            while not stop[0]:
                for _ in range(8):
                    y.copy_(x); x.mul_(1.0000001).add_(y, alpha=1e-9)
                s.synchronize()
    th = threading.Thread(target=worker) if noise else None
    if th: th.start(); time.sleep(0.2)
    for f in range(1, FRAMES + 1):
        ctx.render(cfg["params"]().replace(frames=f, random=g.rand())); out.append(ctx.counters()["candidates"])
    img = ctx.read_image()
    stop[0] = True
    if th: th.join()
    ctx.close(); return out, img
c0, i0 = run(False)
c1, i1 = run(True)
c2, i2 = run(True)
print("kernel", K, "world", WORLD, "frames", FRAMES, "quiet vs noisy survivor counts equal:", c0 == c1, c0 == c2, " images equal:", bool((i0.view(np.uint32) == i1.view(np.uint32)).all()), bool((i0.view(np.uint32) == i2.view(np.uint32)).all()))
