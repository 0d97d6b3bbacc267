"""The CPU oracle against the golden vectors produced by the reference shader on Mesa llvmpipe
(tests/golden/*.npz, generator tests/golden/make_golden.py).  Runs on CPU.

Tolerance: NONE.  The oracle restates the shader in the operation order llvmpipe executes, so the
comparison is bit-for-bit on every channel of every pixel (alpha included); at generation time all
23 cases matched on 100% of pixels and this test keeps it that way."""
import glob
import json
import os

import numpy as np
import pytest

import golden_cases as gc

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASE_FILES = sorted(f for f in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")) if not os.path.basename(f).startswith("probe_"))


def load_case(path, rt):
    z = np.load(path, allow_pickle=False)
    meta = json.loads(str(z["meta"]))
    case = gc.build_cases(rt.scenes)[meta["case"]]
    scene = case["scene"]()
    assert gc.scene_digest(scene) == meta["scene_sha256"], "scene generator drifted from the golden inputs"
    frames = [rt.scenes.FrameParams(**{k: (tuple(v) if isinstance(v, list) else v) for k, v in f.items()}) for f in meta["frames"]]
    return meta, scene, frames, z["expected"]


def test_golden_set_is_present():
    assert len(CASE_FILES) >= 20


@pytest.mark.parametrize("path", CASE_FILES, ids=lambda p: os.path.basename(p)[:-4])
def test_oracle_matches_reference_shader_output(path, rt, oracle):
    meta, scene, frames, expected = load_case(path, rt)
    img = gc.initial_image(meta["init"], meta["width"], meta["height"])
    for p in frames:
        oracle.render(scene, p, img, threads=4)
    neq = (img.view(np.uint32) != expected.view(np.uint32)).any(axis=2)
    assert not neq.any(), f"{int(neq.sum())} of {neq.size} pixels differ from the reference shader's output"


def test_probe_sin_cos_tan(oracle):
    z = np.load(os.path.join(GOLDEN_DIR, "probe_math.npz"))
    s, c = oracle.sincos(z["x"])
    assert (s.view(np.uint32) == z["sin"].view(np.uint32)).all()
    assert (c.view(np.uint32) == z["cos"].view(np.uint32)).all()
    hs, hc = oracle.sincos(z["x"] / np.float32(2))
    assert ((hs / hc).view(np.uint32) == z["tan_half"].view(np.uint32)).all()      # tan = sin / cos, true division


@pytest.mark.parametrize("tag", ["rgba4", "rgb5x"])
def test_probe_cubemap(tag, rt, oracle):
    z = np.load(os.path.join(GOLDEN_DIR, f"probe_cubemap_{tag}.npz"))
    scene = rt.scenes.Scene(env=z["faces"])
    got = oracle.env_lookup(scene, z["dirs"])
    assert (got.view(np.uint32) == z["rgb"].view(np.uint32)).all()


def test_pcg4d_known_answers(oracle):
    # hand-evaluated from the published PCG4D recipe (Jarzynski & Olano, "Hash Functions for GPU Rendering")
    v = np.array([1, 2, 3, 6], np.uint64)
    M = (1 << 32) - 1
    x, y, z, w = [int(t) for t in v]
    x, y, z, w = [(t * 1664525 + 1013904223) & M for t in (x, y, z, w)]
    x = (x + y * w) & M; y = (y + z * x) & M; z = (z + x * y) & M; w = (w + y * z) & M
    x ^= x >> 16; y ^= y >> 16; z ^= z >> 16; w ^= w >> 16
    x = (x + y * w) & M; y = (y + z * x) & M; z = (z + x * y) & M; w = (w + y * z) & M
    assert oracle.pcg4d([1, 2, 3, 6]).tolist() == [x, y, z, w]
