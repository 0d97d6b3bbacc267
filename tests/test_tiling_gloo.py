"""N>1 path on CPU: two processes (gloo) each own the strips rank::2, render them (with the CPU
oracle standing in for the GPU, which this container lacks), run the product's FrameGatherer
(torch.distributed.gather + un-permute) and the result on rank 0 must be bit-identical to a
single-process render of the whole image."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, W, H, strip, frames, out_path, overlap=False):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import raytracer_glsl_amd as rt
    import golden_cases as gc
    from oracle.oracle import CpuOracle
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sc = rt.scenes
    scene = sc.scene_mesh(12, 6, env_size=8)
    orc = CpuOracle()
    gat = rt.tiling.FrameGatherer(W, H, rank, world, torch.device("cpu"), strip)
    rows = rt.tiling.strip_rows_of(H, rank, world, strip)
    full = np.zeros((H, W, 4), np.float32)          # scratch in image order; only own rows are rendered
    result = None
    for p in gc.frame_sequence(sc, sc.params_c2().replace(max_bounce=4), frames):
        for y0 in range(0, len(rows), 8):           # strips are 8-row aligned: render them block by block
            g0 = int(rows[y0])
            y1 = min(g0 + 8, H // 8 * 8)
            if y1 > g0:
                orc.render(scene, p, full, rect=(0, g0, W // 8 * 8, y1), threads=1)
        gat.local[: len(rows)] = torch.from_numpy(full[rows])        # compact local tile buffer
        result = gat.gather(overlap=overlap)                          # the one exchange step per frame
    if overlap:
        result = gat.finish()                                         # overlapped: the last frame's image arrives here
    if rank == 0:
        np.save(out_path, result.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("overlap", [False, True])
def test_two_rank_tiled_render_equals_single(tmp_path, rt, oracle, overlap):
    import golden_cases as gc
    W, H, strip, frames, world = 72, 53, 8, 3 if overlap else 2, 2
    out = str(tmp_path / "gathered.npy")
    port = 29500 + (os.getpid() % 2000) + (17 if overlap else 0)
    mp.spawn(_worker, args=(world, port, W, H, strip, frames, out, overlap), nprocs=world, join=True)
    got = np.load(out)
    sc = rt.scenes
    scene = sc.scene_mesh(12, 6, env_size=8)
    want = np.zeros((H, W, 4), np.float32)
    for p in gc.frame_sequence(sc, sc.params_c2().replace(max_bounce=4), frames):
        oracle.render(scene, p, want, threads=2)
    assert got.shape == want.shape
    assert (got.view(np.uint32) == want.view(np.uint32)).all()
