import sys, os, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import raytracer_glsl_amd as rt
from oracle.oracle import CpuOracle
sc = rt.scenes
o = CpuOracle()
scene = sc.scene_mesh(30, 10, env_size=32)
W, H = 136, 72
for S in (4, 2):
  for cg in (3, 128):
    for nb in (1, 2, 8):
        p = sc.params_c2().replace(frames=1, random=sc.GlibcRand(0).rand(), max_bounce=nb, use_dof=0)
        ctx = rt.host.Context(W, H); ctx.upload_scene(scene)
        ctx.set_option("kernel", 3); ctx.set_option("mf_sets", S); ctx.set_option("mf_chunk_quads", cg); ctx.set_option("counters", 1)
        ctx.render(p); g = ctx.read_image(); cnt = ctx.counters(); ctx.close()
        want = np.zeros((H, W, 4), np.float32); co, _ = o.render(scene, p, want, threads=8)
        neq = (g.view(np.uint32) != want.view(np.uint32)).any(axis=2)
        print("S", S, "chunk_groups", cg, "bounces", nb, "mismatch px", int(neq.sum()), "cand", cnt["candidates"], "tests", cnt["triangle_tests"], "oracle tests", co["triangle_tests"], np.argwhere(neq)[:3].tolist())
