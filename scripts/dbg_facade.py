import sys, os, numpy as np, subprocess, tempfile
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import raytracer_glsl_amd as rt
from test_facade import write_png, OCTAHEDRON, build_demo
from oracle.oracle import CpuOracle
import pathlib
sc = rt.scenes
d = tempfile.mkdtemp()
exe = build_demo(pathlib.Path(d), rt)
open(d + "/mesh.obj", "w").write(OCTAHEDRON)
env = sc.sky_cubemap(16)
for name, face in zip(("right", "left", "top", "bottom", "front", "back"), env): write_png(d + "/" + name + ".png", face)
oracle = CpuOracle()
for frames, reset_at in ((1, 0), (2, 0), (6, 4)):
    W, H = 96, 64
    subprocess.check_call([exe, d, str(W), str(H), str(frames), str(reset_at)], cwd=d, stdout=subprocess.DEVNULL)
    got = np.fromfile(d + "/image.raw", np.float32).reshape(H, W, 4)
    verts = np.fromfile(d + "/vertices.raw", np.float32).reshape(-1, 4)
    scene = sc.Scene(spheres=np.fromfile(d + "/spheres.raw", np.float32).reshape(-1, 8), materials=sc.demo_materials(),
                     meshes=sc.make_meshes([(0, verts.shape[0] // 3, 6)]), vertices=verts,
                     nodes=np.fromfile(d + "/nodes.raw", np.float32).reshape(-1, 12), env=env)
    loop = rt.host.FrameLoop(sc.FrameParams(max_bounce=6))
    want = np.zeros((H, W, 4), np.float32)
    plist = []
    for f in range(1, frames + 1):
        if f == reset_at: loop.reset_buffer()
        p = loop.next_frame(); plist.append(p)
        oracle.render(scene, p, want, threads=4)
    neq = (got.view(np.uint32) != want.view(np.uint32)).any(axis=2)
    print("frames", frames, "reset", reset_at, "mismatching px", int(neq.sum()), "first", np.argwhere(neq)[:5].tolist())
    if neq.any():
        y, x = np.argwhere(neq)[0]; print("  got", got[y, x], "want", want[y, x])
    # same scene through the python binding
    ctx = rt.host.Context(W, H); ctx.upload_scene(scene)
    for p in plist: ctx.render(p)
    py = ctx.read_image(); ctx.close()
    print("   python-binding vs oracle mism:", int((py.view(np.uint32) != want.view(np.uint32)).any(axis=2).sum()),
          " facade vs python-binding mism:", int((py.view(np.uint32) != got.view(np.uint32)).any(axis=2).sum()))
