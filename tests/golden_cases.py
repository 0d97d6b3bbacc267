"""Definitions of the golden-vector cases (shared by the generator tests/golden/make_golden.py and
by the tests that replay them against the oracle and the HIP path).

Each case = a seeded scene (generated, not stored), an image size and a list of per-frame
FrameParams.  The expected output stored in tests/golden/<name>.npz is the RGBA32F image the
REFERENCE SHADER ITSELF produced on Mesa llvmpipe for exactly those inputs.

llvmpipe caps the total number of loop iterations of one shader invocation group at 65535, so every
case keeps n_triangles * bounces well below that; deeper / larger cases are
covered oracle-vs-HIP in tests/test_gpu_parity.py.
"""
from __future__ import annotations

import hashlib
import numpy as np


def scene_digest(scene) -> str:
    h = hashlib.sha256()
    for a in (scene.spheres, scene.materials, scene.meshes, scene.vertices, scene.nodes):
        h.update(np.ascontiguousarray(a).tobytes())
    if scene.env is not None:
        h.update(np.ascontiguousarray(scene.env).tobytes())
    return h.hexdigest()


def frame_sequence(sc, base, n_frames, seed=0, reset_at=(), start_frame=1):
    """Per-frame parameters exactly as the reference's loop produces them: m_frames incremented
    before render (src/window.cpp:42), u_random = rand() per frame (src/renderer.cpp:102), a reset
    uploads the stale frame count and then restarts the count (src/renderer.cpp:98,123-127)."""
    g = sc.GlibcRand(seed)
    out, m_frames = [], start_frame - 1
    for i in range(n_frames):
        m_frames += 1
        reset = (i + 1) in reset_at
        out.append(base.replace(frames=m_frames, random=g.rand(), reset_flag=int(reset)))
        if reset:
            m_frames = 0
    return out


def build_cases(sc):
    """name -> dict(scene=callable, width, height, frames=[FrameParams...], init='zeros'|'ramp')"""
    C = {}
    P1, P2 = sc.params_c1(), sc.params_c2()

    def add(name, scene, w, h, frames, init="zeros"):
        C[name] = dict(scene=scene, width=w, height=h, frames=frames, init=init)

    # --- BASELINE.json configs[0]: 4 spheres, 256x256, 1 spp (ground truth config), plus the lit variant
    add("c1_256", lambda: sc.scene_c1(False), 256, 256, frame_sequence(sc, P1, 1))
    add("c1_light_8f", lambda: sc.scene_c1(True), 96, 96, frame_sequence(sc, P1, 8))
    add("c1_nodof_bounce1", lambda: sc.scene_c1(True), 64, 64, frame_sequence(sc, P1.replace(use_dof=0, max_bounce=1), 2))
    add("c1_bounce16", lambda: sc.scene_c1(True), 64, 64, frame_sequence(sc, P1.replace(max_bounce=16), 2))
    # --- accumulation semantics: frames 1..N, a reset frame with the stale count, then more frames;
    #     and a first frame that averages against a non-zero image
    add("c1_reset_sequence", lambda: sc.scene_c1(True), 64, 64, frame_sequence(sc, P1.replace(use_dof=0), 7, reset_at=(4,)))
    add("c1_preloaded_image", lambda: sc.scene_c1(True), 64, 64, frame_sequence(sc, P1, 2, start_frame=5), init="ramp")
    add("c1_two_samples", lambda: sc.scene_c1(True), 64, 64, frame_sequence(sc, P1.replace(samples=2), 2))
    # --- sizes that are not multiples of 8: the remainder is never written (src/renderer.cpp:132-133)
    add("c1_ragged_70x53", lambda: sc.scene_c1(True), 70, 53, frame_sequence(sc, P1, 2), init="ramp")
    # --- spheres need nodes; node graphs with duplicates and a full stack
    add("spheres_without_nodes", lambda: scene_no_nodes(sc), 64, 64, frame_sequence(sc, P1, 1))
    add("spheres_two_level_tree", lambda: scene_tree(sc), 96, 64, frame_sequence(sc, P1.replace(use_dof=0), 2))
    add("spheres_deep_chain", lambda: scene_chain(sc), 64, 64, frame_sequence(sc, P1.replace(use_dof=0), 1))
    # --- meshes: front / back facing, two overlapping meshes, glass / mirror / unknown material ids
    add("mesh_env_dof", lambda: sc.scene_mesh(20, 10, env_size=32), 128, 128, frame_sequence(sc, P2, 2))
    add("mesh_backfacing", lambda: scene_backfacing(sc), 64, 64, frame_sequence(sc, P2.replace(use_dof=0), 1))
    add("mesh_two_meshes_overlap", lambda: scene_two_meshes(sc), 96, 64, frame_sequence(sc, P2.replace(use_dof=0), 2))
    add("mesh_7k_bounce8", lambda: sc.scene_mesh(70, 50, env_size=64), 96, 64, frame_sequence(sc, P2, 1))
    add("mesh_stacked_duplicates", lambda: scene_stacked(sc), 64, 64, frame_sequence(sc, P2.replace(use_dof=0, max_bounce=4), 2))
    add("mesh_degenerate_inputs", lambda: scene_degenerate(sc), 64, 64, frame_sequence(sc, P2.replace(use_dof=0, max_bounce=4), 2))
    add("zero_bounces_three_samples", lambda: sc.scene_mesh(10, 5, env_size=16), 64, 64,
        frame_sequence(sc, P2.replace(max_bounce=0, samples=3), 2), init="ramp")
    add("mesh_three_samples", lambda: sc.scene_mesh(10, 5, env_size=16), 64, 64, frame_sequence(sc, P2.replace(samples=3, max_bounce=4), 2))
    add("mesh_odd_materials", lambda: scene_odd_materials(sc), 96, 64, frame_sequence(sc, P2.replace(use_dof=0), 2))
    # --- environment: tiny high-contrast cube (face edges, clamp), RGB (3-channel) cube, incomplete cube, background colour
    add("env_noise_cube", lambda: scene_env_only(sc, sc.noise_cubemap(4, 4)), 128, 128, frame_sequence(sc, P2.replace(max_bounce=2), 1))
    add("env_rgb3_cube", lambda: scene_env_only(sc, sc.noise_cubemap(8, 3, seed=11)), 64, 64, frame_sequence(sc, P2.replace(max_bounce=2, use_dof=0), 1))
    add("env_incomplete_cube", lambda: scene_env_only(sc, sc.noise_cubemap(8, 4)[:5]), 64, 64, frame_sequence(sc, P2.replace(max_bounce=2), 1))
    add("env_disabled_background", lambda: sc.scene_mesh(20, 10, env_size=16), 64, 64, frame_sequence(sc, P2.replace(use_envmap=0), 2))
    # --- glass from the inside: camera inside a big glass sphere (total internal reflection path end)
    add("glass_inside_tir", lambda: scene_inside_glass(sc), 96, 96, frame_sequence(sc, P1.replace(max_bounce=8), 2))
    # --- wide aperture depth of field, moved camera (BASELINE.json configs[4] parameters at small size)
    add("dof_wide_c5", lambda: sc.scene_mesh(20, 10, env_size=32), 96, 64, frame_sequence(sc, sc.params_c5(), 2))
    add("camera_moved", lambda: sc.scene_mesh(20, 10, env_size=32), 96, 64,
        frame_sequence(sc, P2.replace(camera_position=(4.0, 3.0, -28.0), camera_forward=(-0.19611613, -0.0, 0.98058068),
                                      camera_right=(-0.98058068, 0.0, -0.19611613), camera_fov=float(sc.radians_f32(50.0))), 2))
    return C


# ---------------------------------------------------------------------------------- extra scenes

def scene_no_nodes(sc):
    s = sc.scene_c1(True)
    s.nodes = np.zeros((0, 12), np.float32)
    return s


def scene_tree(sc):
    """Root with two leaf children whose sphere ranges overlap (duplicates, as KdTree<Sphere>
    produces for straddling primitives, src/kdtree.h:144-154); walk order = right child first."""
    s = sc.scene_c1(True)
    I = sc.INVALID
    box = ((-1e5,) * 3, (1e5,) * 3)
    s.nodes = sc.make_nodes([(box[0], box[1], 1, 2, 0, 0), (box[0], box[1], I, I, 0, 3), (box[0], box[1], I, I, 2, 3)])
    return s


def scene_chain(sc):
    """Seven nodes chained through both children so the 5-entry stack fills up and pushes are
    dropped (shaders/raytracer.glsl:113-121); every node carries one sphere."""
    s = sc.scene_c1(True)
    I = sc.INVALID
    box = ((-1e5,) * 3, (1e5,) * 3)
    items = []
    for k in range(7):
        left = k + 1 if k + 1 < 7 else I
        right = k + 2 if k + 2 < 7 else I
        items.append((box[0], box[1], left, right, k % 5, 1))
    s.nodes = sc.make_nodes(items)
    return s


def scene_backfacing(sc):
    s = sc.scene_mesh(20, 10, env_size=16)
    s.vertices = sc.grid_mesh(20, 10, facing_camera=False)
    return s


def scene_two_meshes(sc):
    """Two mesh ranges over one vertex buffer, the second one re-testing part of the first
    (tie-breaking by visit order) and running past the end of the buffer (clamped)."""
    s = sc.scene_mesh(20, 10, env_size=16)
    n = s.n_triangles
    s.meshes = sc.make_meshes([(0, n // 2 + 30, 0), (n // 2 - 30, n, 0)])
    return s


def scene_odd_materials(sc):
    """Material ids that are out of range, negative, fractional and an unknown material type."""
    s = sc.scene_mesh(24, 10, env_size=16)
    mats = np.array(s.materials)
    extra = sc.make_materials([(sc.rgb(0x3355FF), (0.2, 0.1, 0.0), 0.0, 3)])   # type 3: passes straight through
    s.materials = np.concatenate([mats, extra], axis=0)
    v = np.array(s.vertices)
    w = np.array([0.0, 8.0, 5.9, 42.0, -3.0, 7.5, 8.0, 2.0], np.float32)
    quad = np.repeat(np.arange(v.shape[0] // 6), 6)
    v[:, 3] = w[quad % len(w)]
    s.vertices = v
    return s


def scene_stacked(sc):
    """30 exact copies of one quad in front of a small grid: every ray through the quad has 60
    coincident candidate triangles (equal t: the first visit must win, :349), which also overflows
    the HIP scan's per-lane candidate list and exercises its re-test fallback."""
    s = sc.scene_mesh(10, 5, env_size=16)
    quad = np.array([[-8, -6, 2, 0], [-8, 4, 2, 0], [9, 4, 2, 0], [-8, -6, 2, 7], [9, 4, 2, 7], [9, -6, 2, 7]], np.float32)
    s.vertices = np.concatenate([s.vertices[:150], np.tile(quad, (30, 1)), s.vertices[150:]], axis=0)
    s.meshes = sc.make_meshes([(0, s.vertices.shape[0] // 3, 0)])
    return s


def scene_degenerate(sc):
    """Zero-area triangles, a triangle with a NaN vertex, one with an infinite vertex, one astronomically
    large, one microscopic, a zero-radius sphere and a NaN-radius sphere mixed into a normal scene."""
    s = sc.scene_mesh(10, 5, env_size=16)
    nan, inf = np.float32(np.nan), np.float32(np.inf)
    bad = np.array([
        [0, 0, 2, 0], [0, 0, 2, 0], [0, 0, 2, 0],                     # point triangle
        [-3, 0, 2, 0], [0, 0, 2, 0], [3, 0, 2, 0],                     # collinear
        [-3, 1, 2, 0], [nan, 2, 2, 0], [3, 1, 2, 0],                   # NaN vertex
        [-3, -1, 2, 0], [0, inf, 2, 0], [3, -1, 2, 0],                 # infinite vertex
        [-1e30, -1e30, 3, 2], [-1e30, 1e30, 3, 2], [1e30, 1e30, 3, 2],  # products overflow
        [0, 0, 1, 3], [0, 1e-30, 1, 3], [1e-30, 1e-30, 1, 3],          # denormal-scale
    ], np.float32)
    s.vertices = np.concatenate([s.vertices[:60], bad, s.vertices[60:]], axis=0)
    s.meshes = sc.make_meshes([(0, s.vertices.shape[0] // 3, 0)])
    extra = sc.make_spheres([(2.0, 3.0, -5.0, 0.0, 2), (-4.0, 2.0, -8.0, float("nan"), 3), (5.0, -2.0, -6.0, 1.5, 2)])
    s.spheres = np.concatenate([s.spheres, extra], axis=0)
    s.nodes = sc.single_leaf(len(s.spheres))
    return s


def scene_env_only(sc, env):
    s = sc.scene_c1(True)
    s.env = env
    return s


def scene_inside_glass(sc):
    s = sc.scene_c1(True)
    big = sc.make_spheres([(0.0, 0.0, -35.0, 6.0, 5)])      # glass ball around the default camera
    s.spheres = np.concatenate([s.spheres, big], axis=0)
    s.nodes = sc.single_leaf(len(s.spheres))
    return s


def initial_image(kind: str, w: int, h: int) -> np.ndarray:
    img = np.zeros((h, w, 4), np.float32)
    if kind == "ramp":
        y, x = np.mgrid[0:h, 0:w]
        img[..., 0] = (x % 17) / np.float32(16.0)
        img[..., 1] = (y % 13) / np.float32(12.0)
        img[..., 2] = ((x + y) % 7) / np.float32(6.0)
        img[..., 3] = np.float32(0.25)
    return img
