"""HIP path vs the CPU oracle on the same seeded inputs, through the C ABI.  Bar: bit-exact
(radiance floats, alpha, RNG state), since both sides implement the same IEEE operation order."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


# (kernel, wf_mode, wf_rays): megakernel, wavefront with scalar-fed / LDS-tiled triangle pass
VARIANTS = [(0, 0, 1), (1, 0, 1), (1, 0, 4), (1, 1, 2), (1, 1, 4), (2, 0, 1), (2, 0, 4), (2, 0, 8), (2, 1, 1), (2, 1, 2), (2, 1, 4), (2, 1, 8),
            (4, 1, 3), (4, 4, 3), (4, 2, 1), (4, 16, 32), (4, 1, 32), (4, 32, 32), (4, 8, 5), (4, 32, 32, 2), (4, 1, 3, 2), (4, 4, 32, 0)]   # a 4th entry = option "cull" (2: packet culling on every bounce, 0: off) # kernel 4 = bf16 matrix-core broad phase, one wave per SIMD: (4, quads per group, quads per chunk)


def run_both(rt, oracle, scene, params, W, H, frames=1, rng_state=True, reset_at=None, variant=None):
    sc = rt.scenes
    ctx = rt.host.Context(W, H)
    ctx.upload_scene(scene)
    if variant is not None and variant[0] == 4:
        ctx.set_option("kernel", 4); ctx.set_option("mf_chunk_quads", variant[2]); ctx.set_option("mf_group_quads", variant[1])
        if len(variant) > 3:
            ctx.set_option("cull", variant[3])
    elif variant is not None:
        ctx.set_option("kernel", variant[0]); ctx.set_option("wf_mode", variant[1]); ctx.set_option("wf_rays", variant[2])
        ctx.set_option("wf_chunk", 128)      # small chunks so that even the small test meshes span several work items
        if len(variant) > 3:
            ctx.set_option("wf_early", variant[3])
    if rng_state:
        ctx.set_option("rng_state", 1)
    ctx.set_option("counters", 1)
    img_o = np.zeros((H, W, 4), np.float32)
    g = sc.GlibcRand(0)
    seeds_o = None
    cnt_o = None
    for f in range(1, frames + 1):
        p = params.replace(frames=f, random=g.rand(), reset_flag=int(reset_at == f))
        ctx.render(p)
        cnt_o, seeds_o = oracle.render(scene, p, img_o, threads=8, want_seeds=rng_state)
    img_g = ctx.read_image()
    out = dict(img_g=img_g, img_o=img_o, cnt_g=ctx.counters(), cnt_o=cnt_o)
    if rng_state:
        out["seeds_g"] = ctx.read_rng_state()
        out["seeds_o"] = seeds_o
    ctx.close()
    return out


def assert_bit_exact(r, W, H):
    dw, dh = W // 8 * 8, H // 8 * 8
    a, b = r["img_g"].view(np.uint32), r["img_o"].view(np.uint32)
    neq = (a != b).any(axis=2)
    assert not neq.any(), f"{int(neq.sum())} pixels differ, first at {np.argwhere(neq)[:4].tolist()}"
    if "seeds_g" in r:
        assert (r["seeds_g"][:dh, :dw] == r["seeds_o"][:dh, :dw]).all(), "final PCG4D states differ"
    assert r["cnt_g"]["segments"] == r["cnt_o"]["segments"]
    assert r["cnt_g"]["paths"] == r["cnt_o"]["paths"]
    assert r["cnt_g"]["env_lookups"] == r["cnt_o"]["env_lookups"]


@pytest.mark.parametrize("variant", VARIANTS)
def test_variants_mesh_env(rt, oracle, variant):
    sc = rt.scenes
    r = run_both(rt, oracle, sc.scene_mesh(30, 10, env_size=32), sc.params_c2(), 136, 72, frames=2, variant=variant)
    assert_bit_exact(r, 136, 72)
    assert r["cnt_g"]["triangle_tests"] == r["cnt_o"]["triangle_tests"]


@pytest.mark.parametrize("early", [0, 1, 8])
@pytest.mark.parametrize("variant", [(2, 1, 4), (2, 0, 2), (2, 1, 1)])
def test_wave_level_edge_short_circuit(rt, oracle, variant, early):
    """wf_early = number of leading bounces that use the wave-level short circuit of the three edge tests."""
    sc = rt.scenes
    ctx_opts = variant + (early,)
    r = run_both(rt, oracle, sc.scene_mesh(30, 10, env_size=32), sc.params_c2(), 136, 72, frames=2, variant=ctx_opts)
    assert_bit_exact(r, 136, 72)


@pytest.mark.parametrize("variant", [(0, 0, 1), (1, 1, 2), (2, 1, 4), (2, 0, 8), (4, 2, 3), (4, 4, 32)])
def test_two_samples_per_frame(rt, oracle, variant):
    sc = rt.scenes
    r = run_both(rt, oracle, sc.scene_mesh(20, 10, env_size=16), sc.params_c2().replace(samples=2), 64, 64, frames=2, variant=variant)
    assert_bit_exact(r, 64, 64)


def test_c1_spheres(rt, oracle):
    sc = rt.scenes
    r = run_both(rt, oracle, sc.scene_c1(), sc.params_c1(), 256, 256, frames=2)
    assert_bit_exact(r, 256, 256)


def test_c1_light_no_dof(rt, oracle):
    sc = rt.scenes
    r = run_both(rt, oracle, sc.scene_c1(True), sc.params_c1().replace(use_dof=0), 128, 128, frames=3)
    assert_bit_exact(r, 128, 128)


def test_mesh_env_small(rt, oracle):
    sc = rt.scenes
    r = run_both(rt, oracle, sc.scene_mesh(20, 10, env_size=32), sc.params_c2(), 128, 128, frames=2)
    assert_bit_exact(r, 128, 128)


def test_mesh_10k_crop(rt, oracle):
    sc = rt.scenes
    r = run_both(rt, oracle, sc.scene_mesh(100, 50, env_size=64), sc.params_c2(), 96, 64, frames=1)
    assert_bit_exact(r, 96, 64)
    assert r["cnt_g"]["triangle_tests"] == r["cnt_o"]["triangle_tests"]


def test_triangle_free_scene_defaults_to_the_megakernel_and_explicit_choice_wins(rt, oracle):
    sc = rt.scenes
    scene, p = sc.scene_c1(), sc.params_c1().replace(frames=1, random=sc.GlibcRand(0).rand())
    want = np.zeros((64, 64, 4), np.float32)
    oracle.render(scene, p, want, threads=4)
    for explicit, expect in ((None, 0), (4, 4), (1, 1), (2, 2)):
        ctx = rt.host.Context(64, 64)
        if explicit is not None:
            ctx.set_option("kernel", explicit)
        ctx.upload_scene(scene)
        ctx.render(p)
        assert ctx.get_option("kernel_in_use") == expect
        assert (ctx.read_image().view(np.uint32) == want.view(np.uint32)).all()
        ctx.close()
    mesh = sc.scene_mesh(8, 4, env_size=8)
    ctx = rt.host.Context(64, 64)
    ctx.upload_scene(mesh)
    ctx.render(sc.params_c2().replace(frames=1, random=1))
    assert ctx.get_option("kernel_in_use") == 4
    ctx.close()


def test_camera_ray_keep_bits_are_reused_only_while_they_are_valid(rt, oracle):
    """The camera-ray bounce's keep bits are kept across frames while camera, image and scene stand still (rtgl_amd.hip, `d_keep0`; the
    packet bounds are widened by the depth-of-field jitter).  One context through a sequence that keeps, invalidates and rebuilds them --
    wide aperture, moved camera, depth of field off, a new mesh, an aperture too wide for any bound -- against the oracle after every
    frame, and once more with the reuse switched off (RTGL_AMD_NO_CAMERA_KEEP): the same images."""
    import os
    sc = rt.scenes
    W, H = 296, 184
    scene_a, scene_b = sc.scene_mesh(36, 18, env_size=16), sc.scene_mesh(20, 28, env_size=16)
    base = sc.params_c2().replace(max_bounce=4)
    cams = [dict(camera_aperture=0.5, camera_focal_length=38.0), dict(camera_aperture=0.5, camera_focal_length=38.0), dict(camera_aperture=0.5, camera_focal_length=38.0),
            dict(camera_aperture=0.5, camera_focal_length=38.0, camera_position=(2.0, 1.0, -33.0)), dict(camera_aperture=0.5, camera_focal_length=38.0, camera_position=(2.0, 1.0, -33.0)),
            dict(use_dof=0), dict(use_dof=0), dict(camera_aperture=0.001), "scene_b", dict(camera_aperture=0.001), dict(camera_aperture=0.001),
            dict(camera_aperture=12.0, camera_focal_length=10.0), dict(camera_aperture=12.0, camera_focal_length=10.0)]

    def run(check):
        ctx = rt.host.Context(W, H)
        ctx.upload_scene(scene_a)
        scene = scene_a
        img_o = np.zeros((H, W, 4), np.float32)
        g = sc.GlibcRand(3)
        f = 0
        for step in cams:
            if step == "scene_b":
                ctx.upload_scene(scene_b); scene = scene_b
                continue
            f += 1
            p = base.replace(frames=f, random=g.rand(), **step)
            ctx.render(p)
            if check:
                oracle.render(scene, p, img_o, threads=8)
                got = ctx.read_image()
                assert (got.view(np.uint32) == img_o.view(np.uint32)).all(), f"frame {f} ({step}) differs from the oracle"
        img = ctx.read_image()
        ctx.close()
        return img

    a = run(True)
    os.environ["RTGL_AMD_NO_CAMERA_KEEP"] = "1"
    try:
        b = run(False)
    finally:
        del os.environ["RTGL_AMD_NO_CAMERA_KEEP"]
    assert (a.view(np.uint32) == b.view(np.uint32)).all()


@pytest.mark.parametrize("shape", ["flat", "line", "far", "speck", "nan_vertex", "cube"])
def test_bin_key_cells_on_degenerate_mesh_boxes(rt, oracle, shape):
    """The bin key deals its origin bits to the axes by the extent of the mesh's box and puts the rays that start outside the box into
    cells of their own (rt_wavefront.hpp, ray_bin_key; rtgl_amd.hip, set_bin_cells).  Queue order must never show in a result: meshes
    whose box has no extent on one axis or two, lies far from the origin, is a speck, holds a non-finite vertex, or is a cube -- binned
    queues forced on (`sort_min_rays` 0), three frames, bit for bit against the oracle."""
    sc = rt.scenes
    rng = np.random.default_rng(5)
    n = 400
    c = rng.uniform([-15, -10, 2], [15, 5, 10], size=(n, 3))
    tri = c[:, None, :] + rng.normal(size=(n, 3, 3)) * rng.uniform(0.2, 2.5, n)[:, None, None]
    if shape == "flat": tri[:, :, 2] = 6.0                                   # every vertex in one plane
    elif shape == "line": tri[:, :, 2] = 6.0; tri[:, :, 1] = tri[:, :1, 1] * 0 + np.linspace(-3, -3, n)[:, None]      # ... on one line of it (degenerate triangles)
    elif shape == "far": tri = tri + np.array([3.0e4, -2.0e4, 1.0e4])
    elif shape == "speck": tri = np.array([0.0, -2.0, 5.0]) + (tri - tri.mean((0, 1))) * 1e-3
    elif shape == "cube": tri = rng.uniform(-6, 6, size=(n, 1, 3)) + np.array([0, -4, 12.0]) + rng.normal(size=(n, 3, 3)) * 0.8
    v = np.zeros((n, 3, 4), np.float32); v[..., :3] = tri.astype(np.float32); v[..., 3] = rng.choice([0, 3, 5, 7], size=n)[:, None]
    if shape == "nan_vertex": v[7, 1, 0] = np.nan; v[11, 2, 2] = np.inf
    spheres = sc.demo_spheres(True)
    scene = sc.Scene(spheres=spheres, materials=sc.demo_materials(), meshes=sc.make_meshes([(0, n, 0)]), vertices=v.reshape(-1, 4), nodes=sc.single_leaf(len(spheres)), env=sc.sky_cubemap(16))
    W, H = 200, 120
    cam = dict(camera_position=(3.0e4, -2.0e4 - 2.0, 1.0e4 - 30.0)) if shape == "far" else {}
    p0 = sc.params_c2().replace(max_bounce=6, **cam)
    ctx = rt.host.Context(W, H)
    ctx.set_option("kernel", 4); ctx.set_option("cull", 3); ctx.set_option("sort_min_rays", 0); ctx.set_option("rng_state", 1)
    ctx.upload_scene(scene)
    img_o = np.zeros((H, W, 4), np.float32); g = sc.GlibcRand(9); seeds_o = None
    for f in range(1, 4):
        p = p0.replace(frames=f, random=g.rand())
        ctx.render(p)
        _, seeds_o = oracle.render(scene, p, img_o, threads=8, want_seeds=True)
    img_g = ctx.read_image(); seeds_g = ctx.read_rng_state(); ctx.close()
    neq = (img_g.view(np.uint32) != img_o.view(np.uint32)).any(axis=2)
    assert not neq.any(), f"{shape}: {int(neq.sum())} pixels differ, first at {np.argwhere(neq)[:4].tolist()}"
    assert (seeds_g[:H // 8 * 8, :W // 8 * 8] == seeds_o[:H // 8 * 8, :W // 8 * 8]).all()
