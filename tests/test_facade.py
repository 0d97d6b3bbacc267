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
    defaults = sc.FrameParams()

    def expected(raw, sequence=((1, 0), (2, 0), (3, 0), (4, 1), (1, 0), (2, 0)), start=None):      # renderer.cpp:98,123-127
        used = [rt.host.CFrameParams.from_buffer_copy(raw[i * n:(i + 1) * n]) for i in range(len(sequence))]
        assert [(u.frames, u.reset_flag) for u in used] == list(sequence)
        want = np.zeros((H, W, 4), np.float32) if start is None else start.copy()
        for u in used:
            assert u.max_bounce == 6 and u.samples == 1 and u.use_dof == 1 and u.use_envmap == 1
            assert np.float32(u.camera_fov) == np.float32(defaults.camera_fov) and tuple(u.camera_right) == (-1.0, 0.0, 0.0)
            p = defaults.replace(frames=u.frames, random=u.random, reset_flag=u.reset_flag, max_bounce=u.max_bounce)
            oracle.render(scene, p, want, threads=4)
        return want

    assert (got.view(np.uint32) == expected(raw).view(np.uint32)).all()
    # save_to_file: render_<W>x<H>_<time>_<frames>.png, flipped, 8-bit
    pngs = [f for f in os.listdir(d) if f.startswith(f"render_{W}x{H}_") and f.endswith(".png")]
    assert len(pngs) == 1 and pngs[0].endswith(f"_{frames - reset_at}.png")
    # the same unmodified program with the frame tiled across "three devices" (RTGL_AMD_DEVICES: rtgl_create_multi behind Renderer(w, h),
    # one submit thread per part; the box has one GPU, so all three are device 0)
    d2 = os.path.join(d, "tiled")
    os.mkdir(d2)
    for f in os.listdir(d):
        if f.endswith((".png", ".obj")) and not f.startswith("render_"):
            os.symlink(os.path.join(d, f), os.path.join(d2, f))
    subprocess.check_call([exe, d2, str(W), str(H), str(frames), str(reset_at)], cwd=d2, env=dict(os.environ, RTGL_AMD_DEVICES="0,0,0"))
    tiled = np.fromfile(os.path.join(d2, "image.raw"), np.float32).reshape(H, W, 4)
    assert (tiled.view(np.uint32) == expected(open(os.path.join(d2, "params.raw"), "rb").read()).view(np.uint32)).all()
    # ... and with the library batching four frames per set of launches (RTGL_AMD_FRAME_BATCH, read when the context is created): the reset
    # frame falls inside the first batch, the last two frames are submitted by the read-back
    d3 = os.path.join(d, "batched")
    os.mkdir(d3)
    for f in os.listdir(d):
        if f.endswith((".png", ".obj")) and not f.startswith("render_"):
            os.symlink(os.path.join(d, f), os.path.join(d3, f))
    subprocess.check_call([exe, d3, str(W), str(H), str(frames), str(reset_at)], cwd=d3, env=dict(os.environ, RTGL_AMD_FRAME_BATCH="4"))
    batched = np.fromfile(os.path.join(d3, "image.raw"), np.float32).reshape(H, W, 4)
    assert (batched.view(np.uint32) == expected(open(os.path.join(d3, "params.raw"), "rb").read()).view(np.uint32)).all()
    # lossless outputs and the resumable state.  Run A renders three frames and saves; run B, a new process, loads the state
    # and renders three more: its uniforms continue the frame count (u_frames 4, 5, 6) and its image is the oracle's running mean over
    # A's image -- exactly what an uninterrupted run accumulates
    def fresh(name):
        dn = os.path.join(d, name)
        os.mkdir(dn)
        for f in os.listdir(d):
            if f.endswith((".png", ".obj")) and not f.startswith("render_"):
                os.symlink(os.path.join(d, f), os.path.join(dn, f))
        return dn
    da, db = fresh("run_a"), fresh("run_b")
    state = os.path.join(d, "state.bin")
    subprocess.check_call([exe, da, str(W), str(H), "3", "0"], cwd=da, env=dict(os.environ, RTGL_DEMO_SAVE_STATE=state))
    img_a = np.fromfile(os.path.join(da, "image.raw"), np.float32).reshape(H, W, 4)
    assert (img_a.view(np.uint32) == expected(open(os.path.join(da, "params.raw"), "rb").read(), ((1, 0), (2, 0), (3, 0))).view(np.uint32)).all()
    blob = open(state, "rb").read()
    assert blob[:8] == b"RTGLST01" and struct.unpack("<iii", blob[8:20]) == (W, H, 3) and len(blob) == 24 + W * H * 16
    assert blob[24:] == img_a.tobytes() == open(os.path.join(da, "image_rgba32f.raw"), "rb").read()
    pfm = open(os.path.join(da, "image.pfm"), "rb").read()
    head = f"PF\n{W} {H}\n-1.0\n".encode()
    assert pfm.startswith(head) and pfm[len(head):] == np.ascontiguousarray(img_a[:, :, :3]).tobytes()
    subprocess.check_call([exe, db, str(W), str(H), "3", "0"], cwd=db, env=dict(os.environ, RTGL_DEMO_LOAD_STATE=state))
    img_b = np.fromfile(os.path.join(db, "image.raw"), np.float32).reshape(H, W, 4)
    want_b = expected(open(os.path.join(db, "params.raw"), "rb").read(), ((4, 0), (5, 0), (6, 0)), start=img_a)
    assert (img_b.view(np.uint32) == want_b.view(np.uint32)).all()
    # a state of another size is refused (printed, like the reference's file errors) and the run stops
    assert subprocess.call([exe, fresh("run_c"), str(W + 8), str(H), "1", "0"], cwd=d, env=dict(os.environ, RTGL_DEMO_LOAD_STATE=state)) == 5


REFERENCE_MAIN = "/root/reference/src/main.cpp"


@pytest.mark.skipif(not os.path.exists(REFERENCE_MAIN), reason="the reference checkout exists in the build container only")
def test_reference_main_compiles_unmodified_against_the_facade(tmp_path):
    """the reference's own src/main.cpp, byte for byte, compiles against include/rtgl (the facade's
    renderer.h / window.h / kdtree.h take the place of the reference's headers).  The file is copied to a temporary directory so
    that its sibling headers are out of reach; nlohmann's json.hpp (included, never used by main.cpp) is an empty stand-in there.
    Nothing of the reference is stored in the repo."""
    import shutil
    src = tmp_path / "main.cpp"
    shutil.copyfile(REFERENCE_MAIN, src)
    inc = tmp_path / "inc"
    inc.mkdir()
    (inc / "json.hpp").write_text("")
    subprocess.check_call(["g++", "-std=c++20", "-fsyntax-only", "-I" + os.path.join(ROOT, "include", "rtgl"),
                           "-I" + os.path.join(ROOT, "include"), "-I" + str(inc), str(src)])


def test_boundary_types_stream_like_the_reference(tmp_path):
    """operator<< of KdNode / AABB / Sphere (reference src/kdtree.h:56-60,71-75, src/renderer.h:64-68; used by src/main.cpp:165)."""
    prog = tmp_path / "stream.cpp"
    prog.write_text('#include "rtgl/renderer.h"\n#include <sstream>\n#include <cstdio>\n'
                    'int main() { KdNode n; n.left = 1; n.right = INVALID; n.offset = 3; n.count = 4; Sphere s(glm::vec3(1, 2, 3), 0.5f);\n'
                    ' std::ostringstream a, b, c; a << n; b << s; c << static_cast<const AABB &>(n);\n'
                    ' std::printf("%s\\n%s\\n%s\\n", a.str().c_str(), b.str().c_str(), c.str().c_str()); return 0; }\n')
    exe = str(tmp_path / "stream")
    subprocess.check_call(["g++", "-std=c++17", "-I" + os.path.join(ROOT, "include"), str(prog), "-o", exe, "-fsyntax-only"])
    subprocess.check_call(["g++", "-std=c++17", "-c", "-I" + os.path.join(ROOT, "include"), str(prog), "-o", exe + ".o"])
    src = open(os.path.join(ROOT, "include", "rtgl", "kdtree.h")).read() + open(os.path.join(ROOT, "include", "rtgl", "renderer.h")).read()
    for text in ('"Node { l = "', '", r = "', '", o = "', '", c = "', '"Sphere { c = "', '"AABB { min = "'):
        assert text in src
