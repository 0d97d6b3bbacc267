"""Two processes (one per rank, both on the box's single GPU) render their row strips with the HIP
path, exchange them with the product's FrameGatherer (gloo here, because RCCL refuses two ranks on one
device; the driver's multi-GPU run uses the nccl backend on the same code) and rank 0's assembled image
must be bit-identical to a single-context render."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, W, H, strip, frames, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import raytracer_glsl_amd as rt
    import golden_cases as gc
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sc = rt.scenes
    scene = sc.scene_mesh(40, 20, env_size=32)
    ctx = rt.host.Context(W, H, device=0, rank=rank, world=world, strip_rows=strip)
    ctx.upload_scene(scene)
    gat = rt.tiling.FrameGatherer(W, H, rank, world, torch.device("cpu"), strip)
    result = None
    for p in gc.frame_sequence(sc, sc.params_c2(), frames):
        ctx.render(p)
        gat.local[: ctx.local_rows] = torch.from_numpy(ctx.read_image())
        result = gat.gather()
    if rank == 0:
        np.save(out_path, result.numpy())
    ctx.close()
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_on_one_gpu_match_single_context(tmp_path, rt):
    import golden_cases as gc
    W, H, strip, frames, world = 328, 200, 16, 3, 2
    out = str(tmp_path / "gathered.npy")
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, W, H, strip, frames, out), nprocs=world, join=True)
    got = np.load(out)
    sc = rt.scenes
    scene = sc.scene_mesh(40, 20, env_size=32)
    ctx = rt.host.Context(W, H)
    ctx.upload_scene(scene)
    for p in gc.frame_sequence(sc, sc.params_c2(), frames):
        ctx.render(p)
    want = ctx.read_image()
    ctx.close()
    assert (got.view(np.uint32) == want.view(np.uint32)).all()
