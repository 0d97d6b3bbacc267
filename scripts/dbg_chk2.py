import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
import raytracer_glsl_amd as rt
sc = rt.scenes
W, H = int(os.environ.get("W", "1920")), int(os.environ.get("H", "1080"))
scene = sc.scene_mesh(int(os.environ.get("NX", "100")), int(os.environ.get("NY", "50")), env_size=16)
ctx = rt.host.Context(W, H)
for k, v in (("kernel", 3), ("mf_group_quads", int(os.environ.get("Q", "1"))), ("counters", 1), ("debug_skip_exact", 5)): ctx.set_option(k, v)
ctx.upload_scene(scene)
p = sc.params_c2().replace(frames=1, random=sc.GlibcRand(0).rand(), max_bounce=1, use_dof=int(os.environ.get("DOF", "1")))
for i in range(6):
    ctx.render(p); c = ctx.counters(); print("candidates", c["candidates"], flush=True)
ctx.close()
