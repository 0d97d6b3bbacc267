#!/usr/bin/env python3
"""Times the REFERENCE shader itself on Mesa llvmpipe (build container only: needs Mesa + /root/reference).
Prints Mpaths/s for the C1 configuration and for C2-shaped crops.  llvmpipe stops every loop of a shader
invocation group after 65,535 iterations in total, so the mesh runs are limited to
n_triangles * bounces < 65k; beyond that it does less work than the shader asks for.
    python oracle/time_reference.py"""
import os
import sys
import time

import numpy as np

sys.path[0] = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))   # repo root instead of oracle/
import raytracer_glsl_amd as rt  # noqa: E402
from oracle.oracle import CpuOracle, LlvmpipeReference  # noqa: E402

sc = rt.scenes
RESULTS = []


def run(ref, orc, name, scene, params, W, H, frames):
    img = np.zeros((H, W, 4), np.float32)
    ref.set_scene(scene); ref.set_image(img)
    g = sc.GlibcRand(0)
    ps = [params.replace(frames=f + 1, random=g.rand()) for f in range(frames + 1)]
    ref.render(ps[0])                                   # JIT + first touch, not timed
    t0 = time.perf_counter()
    for p in ps[1:]:
        ref.render(p)
    t_ref = time.perf_counter() - t0
    t0 = time.perf_counter()
    for p in ps[1:]:
        orc.render(scene, p, img, threads=os.cpu_count())
    t_orc = time.perf_counter() - t0
    px = (W // 8 * 8) * (H // 8 * 8) * frames
    RESULTS.append({"case": name, "width": W, "height": H, "frames": frames, "reference_llvmpipe_mpaths_per_s": px / t_ref / 1e6,
                    "oracle_port_mpaths_per_s": px / t_orc / 1e6, "cores": os.cpu_count()})
    print(f"{name:44s} {W}x{H} x{frames}: reference/llvmpipe {px / t_ref / 1e6:8.4f} Mpaths/s   oracle port {px / t_orc / 1e6:8.4f} Mpaths/s   ({os.cpu_count()} cores)")


def main():
    ref, orc = LlvmpipeReference(), CpuOracle()
    print(ref.version)
    run(ref, orc, "C1: 4 spheres, 5 bounces", sc.scene_c1(), sc.params_c1(), 256, 256, 32)
    run(ref, orc, "C1 scene at 1920x1080, 8 bounces", sc.scene_c1(), sc.params_c1().replace(max_bounce=8), 1920, 1080, 2)
    run(ref, orc, "C2 scene, 10k tris, 6 bounces (cap), crop", sc.scene_mesh(100, 50), sc.params_c2().replace(max_bounce=6), 240, 136, 1)
    run(ref, orc, "C2-like, 8k tris, 8 bounces, crop", sc.scene_mesh(80, 50), sc.params_c2(), 240, 136, 1)
    # kept under profiles/ so that bench.py can carry the figure in its JSON line (`cpu_baseline_reference`): the reference cannot
    # travel to the GPU box, so this is the "reference re-timed on host cores" of BASELINE.json, measured where it can run
    import json, platform
    out = {"measured_in": "build container (no GPU)", "cpu": platform.processor() or platform.machine(), "cores": os.cpu_count(),
           "renderer": ref.version, "note": "reference shaders/raytracer.glsl executed unmodified on Mesa llvmpipe; llvmpipe caps the total loop "
           "iterations of a shader invocation at 65,535, so the 10k-triangle scene runs at 6 bounces and the 8-bounce case uses 8k triangles", "cases": RESULTS}
    path = os.path.join(sys.path[0], "profiles", "llvmpipe_reference_timing.json")
    json.dump(out, open(path, "w"), indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
