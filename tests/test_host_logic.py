"""Host-side logic that needs no GPU: frame/reset bookkeeping, glibc rand() restatement, scene
generators, framebuffer tiling arithmetic."""
import ctypes
import numpy as np

import golden_cases as gc


def test_glibc_rand_restatement_matches_libc(rt):
    libc = ctypes.CDLL("libc.so.6")
    for seed in (0, 1, 12345, 2**31 - 1):
        libc.srand(seed)
        g = rt.scenes.GlibcRand(seed)
        assert [libc.rand() for _ in range(200)] == [g.rand() for _ in range(200)]


def test_frame_loop_follows_reference_bookkeeping(rt):
    """src/window.cpp:42 (increment before render) and src/renderer.cpp:98,123-127 (stale count on reset)."""
    loop = rt.host.FrameLoop()
    seq = []
    for i in range(6):
        if i == 3:
            loop.reset_buffer()
        p = loop.next_frame()
        seq.append((p.frames, p.reset_flag))
    assert seq == [(1, 0), (2, 0), (3, 0), (4, 1), (1, 0), (2, 0)]
    want = gc.frame_sequence(rt.scenes, rt.scenes.FrameParams(), 6, reset_at=(4,))
    loop2 = rt.host.FrameLoop()
    got = []
    for i in range(6):
        if i == 3:
            loop2.reset_buffer()
        got.append(loop2.next_frame())
    assert [(p.frames, p.random, p.reset_flag) for p in got] == [(p.frames, p.random, p.reset_flag) for p in want]


def test_default_params_are_the_reference_defaults(rt):
    p = rt.scenes.FrameParams()
    assert (p.max_bounce, p.samples, p.use_dof, p.use_envmap) == (5, 1, 1, 1)          # renderer.h:167-175
    assert p.camera_position == (0.0, 0.0, -35.0) and p.camera_right == (-1.0, 0.0, 0.0)   # renderer.cpp:35, renderer.h:121-123
    assert np.float32(p.camera_fov) == np.float32(33.0) * np.float32(0.017453292519943295)


def test_scene_layout_strides(rt):
    s = rt.scenes.scene_mesh(10, 5, env_size=8)
    assert s.spheres.dtype == np.float32 and s.spheres.shape[1] * 4 == 32
    assert s.materials.shape[1] * 4 == 32 and s.meshes.shape[1] * 4 == 16
    assert s.vertices.shape[1] * 4 == 16 and s.nodes.shape[1] * 4 == 48
    assert s.n_triangles == 100 and s.env.shape == (6, 8, 8, 4)
    # front-facing: geometric normal towards -z (camera side)
    v = s.vertices.reshape(-1, 3, 4)[:, :, :3].astype(np.float64)
    n = np.cross(v[:, 1] - v[:, 0], v[:, 2] - v[:, 0])
    assert (n[:, 2] < 0).all()


def test_scene_generators_are_deterministic(rt):
    a, b = rt.scenes.scene_mesh(30, 20, env_size=16), rt.scenes.scene_mesh(30, 20, env_size=16)
    assert gc.scene_digest(a) == gc.scene_digest(b)
    assert rt.scenes.CONFIGS["C2"]["scene"]().n_triangles == 10000
    assert rt.scenes.CONFIGS["C4"]["scene"]().n_triangles == 100000


def test_strip_partition_covers_image_once(rt):
    t = rt.tiling
    for H, world, strip in ((1080, 8, 16), (1080, 3, 8), (53, 2, 8), (2160, 8, 16), (64, 1, 16)):
        rows = np.concatenate([t.strip_rows_of(H, r, world, strip) for r in range(world)])
        assert sorted(rows.tolist()) == list(range(H))
        perm = t.row_permutation(H, world, strip)
        pad = t.padded_rows(H, world, strip)
        flat = np.full(world * pad, -1)
        for r in range(world):
            rr = t.strip_rows_of(H, r, world, strip)
            flat[r * pad: r * pad + len(rr)] = rr
        assert (flat[perm] == np.arange(H)).all()
