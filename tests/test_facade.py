"""The C++ facade (include/rtgl/renderer.h: the reference's Renderer/Window API over the C ABI).
CPU: it compiles with plain g++ and links librtgl_amd.so.  GPU: a program written in the shape of the
reference's main() renders through it and the result equals the oracle's bit for bit."""
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "raytracer.glsl_amd")


def build_demo(tmp_path, rt):
    rt.host.build_library()
    exe = str(tmp_path / "facade_demo")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "facade_demo.cpp"), "-o", exe,
                           "-L" + PKG, "-lrtgl_amd", "-lz", "-Wl,-rpath," + PKG])
    return exe


def write_png(path, img):
    h, w, c = img.shape
    raw = b"".join(b"\x00" + img[y].tobytes() for y in range(h))

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xFFFFFFFF)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 6 if c == 4 else 2, 0, 0, 0))
                + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


OCTAHEDRON = """# unit octahedron, outward-facing triangles plus one quad to exercise the fan
v 1 0 0
v -1 0 0
v 0 1 0
v 0 -1 0
v 0 0 1
v 0 0 -1
v -2 -1.2 -2
v 2 -1.2 -2
v 2 -1.2 2
v -2 -1.2 2
f 1 3 5
f 3 2 5
f 2 4 5
f 4 1 5
f 3/1/1 1/2/1 6/3/1
f 2//2 3//2 6//2
f 4 2 6
f 1 4 6
f 7 10 9 8
"""


def test_facade_compiles_and_links_with_host_compiler(tmp_path, rt):
    exe = build_demo(tmp_path, rt)
    assert os.path.exists(exe)


@pytest.mark.gpu
def test_facade_program_matches_oracle(tmp_path, rt, oracle):
    sc = rt.scenes
    exe = build_demo(tmp_path, rt)
    d = str(tmp_path)
    open(os.path.join(d, "mesh.obj"), "w").write(OCTAHEDRON)
    env = sc.sky_cubemap(16)
    for name, face in zip(("right", "left", "top", "bottom", "front", "back"), env):
        write_png(os.path.join(d, name + ".png"), face)
    W, H, frames, reset_at = 96, 64, 6, 4
    subprocess.check_call([exe, d, str(W), str(H), str(frames), str(reset_at), "save"], cwd=d)
    got = np.fromfile(os.path.join(d, "image.raw"), np.float32).reshape(H, W, 4)
    verts = np.fromfile(os.path.join(d, "vertices.raw"), np.float32).reshape(-1, 4)
    assert verts.shape[0] == 3 * 10 and (verts[:, 3] == 6).all()          # 8 triangles + a fanned quad
    scene = sc.Scene(spheres=np.fromfile(os.path.join(d, "spheres.raw"), np.float32).reshape(-1, 8),
                     materials=sc.demo_materials(), meshes=sc.make_meshes([(0, verts.shape[0] // 3, 6)]), vertices=verts,
                     nodes=np.fromfile(os.path.join(d, "nodes.raw"), np.float32).reshape(-1, 12), env=env)
    # the uniforms the facade uploaded each frame (u_random is libc rand(), whose stream other libraries in the
    # process may also draw from, so the values are taken from the program rather than assumed)
    import ctypes
    raw = open(os.path.join(d, "params.raw"), "rb").read()
    n = ctypes.sizeof(rt.host.CFrameParams)
    used = [rt.host.CFrameParams.from_buffer_copy(raw[i * n:(i + 1) * n]) for i in range(frames)]
    assert [(u.frames, u.reset_flag) for u in used] == [(1, 0), (2, 0), (3, 0), (4, 1), (1, 0), (2, 0)]   # renderer.cpp:98,123-127
    defaults = sc.FrameParams()
    want = np.zeros((H, W, 4), np.float32)
    for u in used:
        assert u.max_bounce == 6 and u.samples == 1 and u.use_dof == 1 and u.use_envmap == 1
        assert np.float32(u.camera_fov) == np.float32(defaults.camera_fov) and tuple(u.camera_right) == (-1.0, 0.0, 0.0)
        p = defaults.replace(frames=u.frames, random=u.random, reset_flag=u.reset_flag, max_bounce=u.max_bounce)
        oracle.render(scene, p, want, threads=4)
    assert (got.view(np.uint32) == want.view(np.uint32)).all()
    # save_to_file: render_<W>x<H>_<time>_<frames>.png, flipped, 8-bit
    pngs = [f for f in os.listdir(d) if f.startswith(f"render_{W}x{H}_") and f.endswith(".png")]
    assert len(pngs) == 1 and pngs[0].endswith(f"_{frames - reset_at}.png")
