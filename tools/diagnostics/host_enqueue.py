# host time of rtgl_render_frame (enqueue only) beside the device time of the frame: what ONE thread driving N devices can sustain
# usage: python tools/diagnostics/host_enqueue.py [world] [frames per burst] [multi]
#   multi: ONE rtgl_create_multi context of `world` parts (all on device 0 here), submitted by one thread per part
#          (RTGL_AMD_MULTI_THREADS=0: by the calling thread, part after part) instead of rank 0 of `world`
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import raytracer_glsl_amd as rt
sc = rt.scenes
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
burst = int(sys.argv[2]) if len(sys.argv) > 2 else 8
cfg = sc.CONFIGS[os.environ.get("CONFIG", "C2")]; W, H = cfg["width"], cfg["height"]; scene = cfg["scene"](); base = cfg["params"]()
multi = len(sys.argv) > 3 and sys.argv[3] == "multi"
ctx = rt.host.Context(W, H, devices=[0] * world, strip_rows=8) if multi else rt.host.Context(W, H, device=0, rank=0, world=world, strip_rows=8)
ctx.upload_scene(scene)
g = sc.GlibcRand(0); ps = [base.replace(frames=f, random=g.rand()) for f in range(1, 400)]
for p in ps[:20]: ctx.render(p, sync=False)
ctx.synchronize()
host, dev, k = [], [], 20
for _ in range(10):
    t0 = time.perf_counter()
    for p in ps[k:k + burst]: ctx.render(p, sync=False)
    t1 = time.perf_counter(); ctx.synchronize(); t2 = time.perf_counter(); k += burst
    host.append((t1 - t0) / burst); dev.append((t2 - t0) / burst)
host.sort(); dev.sort()
print(f"world {world}{' (one multi-device context)' if multi else ''}: enqueue {host[len(host) // 2] * 1e6:.0f} us per frame on the host (median of 10 bursts of {burst}), burst done in {dev[len(dev) // 2] * 1e3:.3f} ms per frame", flush=True)
ctx.close()
