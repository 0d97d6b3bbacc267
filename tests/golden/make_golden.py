#!/usr/bin/env python3
"""Generates tests/golden/*.npz by executing the REFERENCE's own compute shader
(/root/reference/shaders/raytracer.glsl, read at run time, never copied) on Mesa llvmpipe through
oracle/_ref/liblpgl.so.  Only runs in the build container (needs Mesa's swrast_dri.so and the
reference checkout).  Each .npz holds inputs-by-recipe (case name -> tests/golden_cases.py, plus a
SHA-256 of the generated scene bytes and the per-frame uniforms) and the expected RGBA32F image.

    python tests/golden/make_golden.py [case ...]
"""
import dataclasses
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import raytracer_glsl_amd as rt  # noqa: E402
import golden_cases as gc  # noqa: E402
from oracle.oracle import CpuOracle, LlvmpipeReference  # noqa: E402


def main():
    sc = rt.scenes
    cases = gc.build_cases(sc)
    names = sys.argv[1:] or list(cases)
    ref = LlvmpipeReference()
    orc = CpuOracle()
    print("reference runs on:", ref.version)
    for name in names:
        c = cases[name]
        scene = c["scene"]()
        W, H = c["width"], c["height"]
        img0 = gc.initial_image(c["init"], W, H)
        ref.set_scene(scene)
        ref.set_image(img0)
        img_o = img0.copy()
        for p in c["frames"]:
            ref.render(p)
            orc.render(scene, p, img_o, threads=8)
        expected = ref.read_image()
        same = (expected.view(np.uint32) == img_o.view(np.uint32)).all(axis=2)
        meta = dict(case=name, width=W, height=H, init=c["init"], scene_sha256=gc.scene_digest(scene),
                    frames=[dataclasses.asdict(p) for p in c["frames"]], generated_on=ref.version,
                    oracle_bit_identical_pixels=int(same.sum()), pixels=int(same.size))
        np.savez_compressed(os.path.join(HERE, name + ".npz"), expected=expected, meta=np.array(json.dumps(meta)))
        print(f"{name:28s} {W}x{H} frames={len(c['frames'])} oracle bit-identical {same.mean() * 100:.4f}%"
              f"  mean={expected[..., :3].mean():.4f}")


if __name__ == "__main__":
    main()
