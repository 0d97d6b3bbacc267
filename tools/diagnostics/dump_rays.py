# Diagnostics: the rays entering bounce b of one frame of a named configuration, traced by the CPU oracle (oracle_set_ray_dump), as
# /tmp/rays/<config>_b<b>.npy: (pixels, 6) float32 = o.xyz d.xyz, NaN where the path ended earlier.  Input of cull_emulation.py.
# usage: python tools/diagnostics/dump_rays.py C2 1 [threads]
import ctypes as C, sys, time, numpy as np
sys.path.insert(0, "/root/repo")
import raytracer_glsl_amd as rt
from oracle.oracle import CpuOracle
name, bounce = sys.argv[1], int(sys.argv[2]); threads = int(sys.argv[3]) if len(sys.argv) > 3 else 8
cfg = rt.scenes.CONFIGS[name]; W, H = cfg["width"], cfg["height"]
scene = cfg["scene"](); p = cfg["params"]().replace(frames=1, random=rt.scenes.GlibcRand(0).rand())
orc = CpuOracle()
buf = np.full((H * W, 6), np.nan, np.float32)
orc.lib.oracle_set_ray_dump(buf.ctypes.data_as(C.c_void_p), C.c_uint32(bounce))
img = np.zeros((H, W, 4), np.float32)
t0 = time.time(); orc.render(scene, p.replace(max_bounce=bounce + 1), img, threads=threads)
orc.lib.oracle_set_ray_dump(None, C.c_uint32(0))
live = ~np.isnan(buf[:, 0])
print(name, "bounce", bounce, "live rays", int(live.sum()), "of", H * W, "in", round(time.time() - t0, 1), "s")
np.save(f"/tmp/rays/{name}_b{bounce}.npy", buf)
