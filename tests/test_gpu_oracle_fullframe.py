"""Oracle parity at BASELINE.json's full sizes: the HIP path against the CPU oracle -- itself pinned bit for bit to
the reference shader's llvmpipe output (tests/test_oracle_golden.py) -- on

  * the WHOLE C2 frame (1920 x 1080, 8 bounces, 10,000 triangles), two accumulated frames: image, alpha, the final PCG4D state of every
    pixel, and the work counters (paths, segments, triangle tests, environment lookups);
  * >= 10 % of C4 (100,000 triangles) and of C5 (3840 x 2160, 16 bounces, aperture 0.5) as 8-row strips uniformly strided over the
    frame (sky rows, the rows that hold the spheres and the glass, both halves of the depth-of-field blur): image and RNG state.

Tolerance: none.  The variant-against-variant tests of test_gpu_fullsize.py say the kernels agree with each other; this file says they
agree with the reference's arithmetic."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def gpu_frames(rt, cfg_name, frames, counters=False):
    sc = rt.scenes
    cfg = sc.CONFIGS[cfg_name]
    W, H = cfg["width"], cfg["height"]
    scene = cfg["scene"]()
    ctx = rt.host.Context(W, H)
    ctx.set_option("rng_state", 1)
    if counters:
        ctx.set_option("counters", 1)
    ctx.upload_scene(scene)
    g = sc.GlibcRand(0)
    plist, cnts = [], []
    for f in range(1, frames + 1):
        p = cfg["params"]().replace(frames=f, random=g.rand())
        ctx.render(p)
        plist.append(p)
        if counters:
            cnts.append(ctx.counters())
    img, seeds = ctx.read_image(), ctx.read_rng_state()
    ctx.close()
    return img, seeds, cnts, scene, plist


def test_c2_full_frame_matches_oracle(rt, oracle):
    img, seeds, cnts, scene, plist = gpu_frames(rt, "C2", 2, counters=True)
    H, W = img.shape[:2]
    want = np.zeros_like(img)
    want_seeds = None
    for p, cnt in zip(plist, cnts):
        oc, want_seeds = oracle.render(scene, p, want, threads=16, want_seeds=True)
        assert cnt["paths"] == oc["paths"] == W * H
        assert cnt["segments"] == oc["segments"]
        assert cnt["triangle_tests"] == oc["triangle_tests"]
        assert cnt["env_lookups"] == oc["env_lookups"]
    neq = (img.view(np.uint32) != want.view(np.uint32)).any(axis=2)
    assert not neq.any(), f"{int(neq.sum())} of {neq.size} pixels differ from the oracle, first at {list(zip(*np.nonzero(neq)))[:8]}"
    sneq = (seeds.reshape(H, W, 4) != want_seeds).any(axis=2)
    assert not sneq.any(), f"{int(sneq.sum())} final RNG states differ"


@pytest.mark.parametrize("cfg_name,n_strips", [("C4", 14), ("C5", 28)])
def test_strided_tenth_of_the_frame_matches_oracle(cfg_name, n_strips, rt, oracle):
    img, seeds, _, scene, plist = gpu_frames(rt, cfg_name, 1)
    H, W = img.shape[:2]
    seeds = seeds.reshape(H, W, 4)
    strips = [int(round(i * (H - 8) / (n_strips - 1))) // 8 * 8 for i in range(n_strips)]
    assert len(set(strips)) == n_strips and 8 * n_strips >= 0.10 * H
    want = np.zeros_like(img)
    for y0 in strips:
        _, ws = oracle.render(scene, plist[0], want, rect=(0, y0, W, y0 + 8), threads=16, want_seeds=True)
        assert (img[y0:y0 + 8].view(np.uint32) == want[y0:y0 + 8].view(np.uint32)).all(), f"{cfg_name}: strip at row {y0}"
        assert (seeds[y0:y0 + 8] == ws[y0:y0 + 8]).all(), f"{cfg_name}: RNG states of the strip at row {y0}"
    assert (img[..., 3] == 1.0).all()
