"""HIP path, through the C ABI, against the golden vectors the REFERENCE SHADER produced on Mesa
llvmpipe (tests/golden/).  Tolerance: none -- bit-for-bit on all four channels of every pixel."""
import os

import numpy as np
import pytest

import golden_cases as gc
from test_oracle_golden import CASE_FILES, load_case

pytestmark = pytest.mark.gpu


def render_case(rt, meta, scene, frames, options=()):
    W, H = meta["width"], meta["height"]
    ctx = rt.host.Context(W, H)
    for k, v in options:
        ctx.set_option(k, v)
    ctx.upload_scene(scene)
    ctx.write_image(gc.initial_image(meta["init"], W, H))
    for p in frames:
        ctx.render(p)
    img = ctx.read_image()
    ctx.close()
    return img


@pytest.mark.parametrize("path", CASE_FILES, ids=lambda p: os.path.basename(p)[:-4])
def test_default_kernel_matches_reference_shader_output(path, rt):
    """library defaults (kernel 4: one wave per SIMD, 32 quads per group) -- every golden case, bit for bit."""
    meta, scene, frames, expected = load_case(path, rt)
    img = render_case(rt, meta, scene, frames)
    neq = (img.view(np.uint32) != expected.view(np.uint32)).any(axis=2)
    assert not neq.any(), f"{int(neq.sum())} of {neq.size} pixels differ from the reference shader's output"


@pytest.mark.parametrize("path", CASE_FILES, ids=lambda p: os.path.basename(p)[:-4])
def test_matrix_core_broad_phase_small_chunks_matches_reference_shader_output(path, rt):
    """kernel 4 with chunks and groups small enough that the small golden meshes span several of each (a chunk of 3 quads cuts
    groups of 2 in the middle: segments of one and two quads, odd and even tile counts)."""
    meta, scene, frames, expected = load_case(path, rt)
    img = render_case(rt, meta, scene, frames, options=(("kernel", 4), ("mf_chunk_quads", 3), ("mf_group_quads", 2), ("cull", 0)))
    neq = (img.view(np.uint32) != expected.view(np.uint32)).any(axis=2)
    assert not neq.any(), f"{int(neq.sum())} of {neq.size} pixels differ from the reference shader's output"


@pytest.mark.parametrize("path", CASE_FILES, ids=lambda p: os.path.basename(p)[:-4])
def test_one_wave_per_simd_scan_matches_reference_shader_output(path, rt):
    """kernel 4 with one-quad groups and one-quad chunks (every segment is a prologue, one trip and an epilogue) -- every golden case."""
    meta, scene, frames, expected = load_case(path, rt)
    img = render_case(rt, meta, scene, frames, options=(("kernel", 4), ("mf_chunk_quads", 1), ("mf_group_quads", 1), ("cull", 2)))
    assert (img.view(np.uint32) == expected.view(np.uint32)).all()


@pytest.mark.parametrize("waves", [1, 2])
@pytest.mark.parametrize("path", CASE_FILES, ids=lambda p: os.path.basename(p)[:-4])
def test_dynamic_work_distribution_matches_reference_shader_output(path, waves, rt):
    """kernel 4 with its items claimed from counters (rt_scan.hpp, `scan_dynamic` = 2: the default only from 41k triangles on), chunks of
    two quads so that the small golden meshes have several chunks for the blocks to move between, the cull on every bounce (its
    compacted item lists), one and two waves per SIMD -- every golden case."""
    meta, scene, frames, expected = load_case(path, rt)
    img = render_case(rt, meta, scene, frames, options=(("kernel", 4), ("scan_dynamic", 2), ("scan_waves", waves), ("mf_chunk_quads", 2), ("cull", 2)))
    assert (img.view(np.uint32) == expected.view(np.uint32)).all()


@pytest.mark.parametrize("dyn", [1, 2])
@pytest.mark.parametrize("path", CASE_FILES, ids=lambda p: os.path.basename(p)[:-4])
def test_binned_queues_match_reference_shader_output(path, dyn, rt):
    """kernel 4 with every queue binned by (direction cell, origin cell) before it is scanned and culled (`cull` = 3 with
    `sort_min_rays` = 0: the default bins only queues of 131,072 rays and more), static and dynamic work distribution, chunks of two
    quads -- every golden case.  Queue order must never show in a result."""
    meta, scene, frames, expected = load_case(path, rt)
    img = render_case(rt, meta, scene, frames, options=(("kernel", 4), ("cull", 3), ("sort_min_rays", 0), ("scan_dynamic", dyn), ("mf_chunk_quads", 2)))
    assert (img.view(np.uint32) == expected.view(np.uint32)).all()


@pytest.mark.parametrize("path", CASE_FILES, ids=lambda p: os.path.basename(p)[:-4])
def test_fp32_scan_kernel_matches_reference_shader_output(path, rt):
    """kernel 2: fp32 VALU filter + exact test (the variant without matrix cores) -- every golden case, bit for bit."""
    meta, scene, frames, expected = load_case(path, rt)
    img = render_case(rt, meta, scene, frames, options=(("kernel", 2), ("wf_chunk", 128)))
    assert (img.view(np.uint32) == expected.view(np.uint32)).all()


@pytest.mark.parametrize("kernel", [0, 1])
@pytest.mark.parametrize("name", ["c1_reset_sequence", "c1_two_samples", "mesh_two_meshes_overlap", "mesh_odd_materials",
                                  "glass_inside_tir", "c1_ragged_70x53", "spheres_deep_chain", "env_noise_cube"])
def test_other_kernels_match_reference_shader_output(name, kernel, rt):
    path = [p for p in CASE_FILES if os.path.basename(p) == name + ".npz"][0]
    meta, scene, frames, expected = load_case(path, rt)
    img = render_case(rt, meta, scene, frames, options=(("kernel", kernel),))
    assert (img.view(np.uint32) == expected.view(np.uint32)).all()


def test_u8_readback_matches_gl_conversion(rt):
    """rtgl_read_image_u8 = clamp, *255, round-to-nearest (GL_UNSIGNED_BYTE readback, renderer.cpp:223)."""
    path = [p for p in CASE_FILES if os.path.basename(p) == "c1_light_8f.npz"][0]
    meta, scene, frames, expected = load_case(path, rt)
    W, H = meta["width"], meta["height"]
    ctx = rt.host.Context(W, H)
    ctx.upload_scene(scene)
    for p in frames:
        ctx.render(p)
    f32 = ctx.read_image()
    u8 = ctx.read_image_u8(flip=False)
    u8f = ctx.read_image_u8(flip=True)
    ctx.close()
    want = np.rint(np.clip(f32, 0.0, 1.0) * np.float32(255.0)).astype(np.uint8)
    assert (u8 == want).all()
    assert (u8f == want[::-1]).all()
    assert want.max() == 255 and (f32 > 1.0).any()      # the light saturates: the clamp is exercised
