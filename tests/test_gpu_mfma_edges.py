"""Edge cases for the conservative filters in front of the exact triangle test: the result must stay
bit-identical to the oracle whatever the broad phase makes of the input -- non-finite and huge vertices (bounds become
NaN: nothing may be rejected), degenerate triangles, geometry far from the world origin (cancellation in the Pluecker
form), needle triangles, many coplanar duplicates (survivor queue drains in mid-step) and rays starting on the surface."""
import numpy as np
import pytest

from test_gpu_parity import assert_bit_exact, run_both

pytestmark = pytest.mark.gpu

SCANS = [(4, 16, 32, 2), (4, 1, 3), (4, 32, 32), (2, 1, 4), (0, 0, 1), (4, 1, 1, 2), (4, 4, 5)]      # default and small groups / chunks of the matrix scan, fp32 scan, megakernel


def scene_with(rt, vertices, spheres=True, env=32):
    sc = rt.scenes
    sp = sc.demo_spheres(True) if spheres else np.zeros((0, 8), np.float32)
    v = np.ascontiguousarray(vertices, np.float32).reshape(-1, 4)
    return sc.Scene(spheres=sp, materials=sc.demo_materials(), meshes=sc.make_meshes([(0, v.shape[0] // 3, 0)]), vertices=v,
                    nodes=sc.single_leaf(len(sp)) if len(sp) else np.zeros((0, 12), np.float32), env=sc.sky_cubemap(env))


@pytest.mark.parametrize("variant", SCANS)
def test_non_finite_huge_and_degenerate_vertices(rt, oracle, variant):
    sc = rt.scenes
    v = sc.grid_mesh(24, 12).reshape(-1, 3, 4).copy()
    rng = np.random.default_rng(5)
    bad = rng.choice(v.shape[0], 40, replace=False)
    v[bad[0:6], 1, 0] = np.nan                     # NaN coordinate
    v[bad[6:12], 2, 2] = np.inf                    # infinite coordinate
    v[bad[12:18], 0, :3] = 3.0e19                  # beyond the 1e18 guard of the bounds
    v[bad[18:24], 0, :3] *= 1.0e6                  # large but legal: a needle through the scene
    v[bad[24:30], 1] = v[bad[24:30], 0]            # zero-area: two equal vertices
    v[bad[30:36], 1:, :3] = v[bad[30:36], :1, :3]  # a point
    v[bad[36:40], :, :3] *= 1.0e-7                 # microscopic, near the origin
    r = run_both(rt, oracle, scene_with(rt, v), sc.params_c2(), 120, 72, frames=2, variant=variant)
    assert_bit_exact(r, 120, 72)


@pytest.mark.parametrize("variant", SCANS[:4])
@pytest.mark.parametrize("shift", [(1000.0, -2000.0, 500.0), (-3.0e4, 1.0e4, 2.0e4)])
def test_scene_far_from_the_world_origin(rt, oracle, variant, shift):
    """Everything -- mesh, spheres, camera -- translated: |o| and |v| are thousands of edge lengths, which is where the
    reference's own Pluecker evaluation gets noisy and the margins must follow it."""
    sc = rt.scenes
    off = np.asarray(shift, np.float32)
    v = sc.grid_mesh(30, 10).reshape(-1, 4).copy()
    v[:, :3] += off
    scene = scene_with(rt, v)
    scene.spheres[:, :3] += off
    p = sc.params_c2()
    p = p.replace(camera_position=tuple(np.asarray(p.camera_position, np.float32) + off))
    r = run_both(rt, oracle, scene, p, 120, 72, frames=2, variant=variant)
    assert_bit_exact(r, 120, 72)


@pytest.mark.parametrize("variant", SCANS[:4])
def test_stacked_coplanar_duplicates_drain_the_survivor_queue(rt, oracle, variant):
    """The same patch 60 times over: every ray that hits it has 60+ survivors in consecutive tiles, so the per-wave queue
    fills and is drained between half steps; ties in t must resolve to the first visited triangle (:349)."""
    sc = rt.scenes
    patch = sc.grid_mesh(4, 3, x0=-12.0, x1=12.0, y0=-10.0, y1=4.0, amp=0.0).reshape(-1, 4)
    v = np.concatenate([patch] * 60 + [sc.grid_mesh(10, 6).reshape(-1, 4)])
    r = run_both(rt, oracle, scene_with(rt, v), sc.params_c2(), 136, 72, frames=2, variant=variant)
    assert_bit_exact(r, 136, 72)
    assert r["cnt_g"]["candidates"] > 60 * 136 * 72 // 4


@pytest.mark.parametrize("variant", SCANS[:4])
def test_needles_and_mixed_scales_in_one_group(rt, oracle, variant):
    """Triangles of very different size share a group, so E, Ml, P, Pw are dominated by the largest: the bound must still hold
    for the small ones."""
    sc = rt.scenes
    rng = np.random.default_rng(11)
    v = sc.grid_mesh(20, 10).reshape(-1, 3, 4).copy()
    idx = rng.choice(v.shape[0], 60, replace=False)
    cen = v[idx].mean(axis=1, keepdims=True)
    v[idx[:30], :, :3] = cen[:30, :, :3] + (v[idx[:30], :, :3] - cen[:30, :, :3]) * 1.0e-3        # tiny
    v[idx[30:], 0, :3] += rng.normal(size=(30, 3)).astype(np.float32) * 40.0                      # long needles
    r = run_both(rt, oracle, scene_with(rt, v), sc.params_c2(), 120, 72, frames=2, variant=variant)
    assert_bit_exact(r, 120, 72)


def test_candidate_buffer_overflow_falls_back_to_in_place_exact_tests(rt, oracle, monkeypatch):
    """kernel 4 appends survivors to a global buffer for the narrow-phase kernel; pairs that do not fit are tested inside the scan.
    With the buffer clamped to 1000 pairs almost everything takes that path: the result must not change."""
    monkeypatch.setenv("RTGL_DEBUG_CAND_CAP", "4")
    sc = rt.scenes
    patch = sc.grid_mesh(4, 3, x0=-12.0, x1=12.0, y0=-10.0, y1=4.0, amp=0.0).reshape(-1, 4)
    v = np.concatenate([patch] * 20 + [sc.grid_mesh(20, 10).reshape(-1, 4)])
    r = run_both(rt, oracle, scene_with(rt, v), sc.params_c2(), 136, 72, frames=2, variant=(4, 4, 32))
    assert_bit_exact(r, 136, 72)
    assert r["cnt_g"]["candidates"] > 20000
