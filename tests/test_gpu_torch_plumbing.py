"""The plumbing bench.py relies on: rendering straight into a torch tensor (rtgl_bind_device_image) on
torch's current stream (rtgl_set_stream), the single-rank FrameGatherer path and the accumulated HIP-event
timing.  The image must equal what the context's own buffer/stream path produces."""
import numpy as np
import pytest
import torch

import golden_cases as gc

pytestmark = pytest.mark.gpu


def test_render_into_torch_tensor_on_torch_stream(rt):
    sc = rt.scenes
    W, H = 200, 120
    scene = sc.scene_mesh(30, 12, env_size=16)
    frames = gc.frame_sequence(sc, sc.params_c2(), 3)
    ref = rt.host.Context(W, H)
    ref.upload_scene(scene)
    for p in frames:
        ref.render(p)
    want = ref.read_image()
    ref.close()

    dev = torch.device("cuda", 0)
    ctx = rt.host.Context(W, H, device=0, rank=0, world=1, strip_rows=16)
    ctx.upload_scene(scene)
    gat = rt.tiling.FrameGatherer(W, H, 0, 1, dev, 16)
    side = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(side):
        ctx.bind_device_image(gat.local.data_ptr())
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        ctx.set_option("kernel_timing", 1)
        ctx.timing_reset()
        full = None
        for p in frames:
            ctx.render(p, sync=False)          # queued back to back, no host sync
            full = gat.gather()
        side.synchronize()
        t = ctx.accumulated_timing()
    assert t["frames"] == 3 and t["intersect_launches"] == 3 * frames[0].max_bounce
    assert 0.0 < t["intersect_ms"] <= t["frame_ms"]
    got = full.cpu().numpy()
    assert (got.view(np.uint32) == want.view(np.uint32)).all()
    # sampled timing: with period 2 every other frame carries the event pairs
    with torch.cuda.stream(side):
        ctx.set_option("kernel_timing", 2)
        ctx.timing_reset()
        for p in gc.frame_sequence(sc, sc.params_c2(), 5):
            ctx.render(p, sync=False)
        side.synchronize()
        t2 = ctx.accumulated_timing()
    assert t2["frames"] == 3 and t2["intersect_launches"] == 3 * frames[0].max_bounce
    # detach again before the tensor goes away
    ctx.bind_device_image(0)
    ctx.set_stream(0)
    ctx.close()


def _one_rank_rccl_worker(port, out_path):
    """Everything bench.py does with torch.distributed for N > 1 -- nccl (= RCCL) process group bound to the device, barrier,
    gather of the tile buffer into rank 0's list, index_select un-permute, all_reduce of the timing scalar -- on a group of ONE rank."""
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    import raytracer_glsl_amd as rt
    import golden_cases as gc2
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    sc = rt.scenes
    W, H = 200, 120
    ctx = rt.host.Context(W, H, device=0, rank=0, world=1, strip_rows=16)
    ctx.upload_scene(sc.scene_mesh(30, 12, env_size=16))
    gat = rt.tiling.FrameGatherer(W, H, 0, 1, dev, 16, force_collective=True)
    ctx.bind_device_image(gat.local.data_ptr())
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    dist.barrier(); torch.cuda.synchronize()
    full = None
    frames = gc2.frame_sequence(sc, sc.params_c2(), 4)
    ctx.render(frames[0], sync=False)
    full = gat.gather()                          # synchronous exchange
    for p in frames[1:]:
        ctx.render(p, sync=False)
        gat.gather(overlap=True)                 # snapshot + async gather, next frame renders meanwhile (as bench.py does for N > 1)
    full = gat.finish()
    dist.barrier(); torch.cuda.synchronize()
    tt = torch.tensor([1.5], dtype=torch.float64, device=dev)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    assert float(tt.item()) == 1.5
    np.save(out_path, full.cpu().numpy())
    ctx.bind_device_image(0); ctx.set_stream(0); ctx.close()
    dist.destroy_process_group()


def test_rccl_gather_path_on_a_one_rank_group(tmp_path, rt):
    import os
    import torch.multiprocessing as mp
    out = str(tmp_path / "full.npy")
    ctxm = mp.get_context("spawn")
    proc = ctxm.Process(target=_one_rank_rccl_worker, args=(29700 + os.getpid() % 2000, out))
    proc.start(); proc.join(300)
    assert proc.exitcode == 0
    sc = rt.scenes
    ref = rt.host.Context(200, 120)
    ref.upload_scene(sc.scene_mesh(30, 12, env_size=16))
    for p in gc.frame_sequence(sc, sc.params_c2(), 4):
        ref.render(p)
    want = ref.read_image(); ref.close()
    assert (np.load(out).view(np.uint32) == want.view(np.uint32)).all()


def test_set_stream_refuses_a_second_pipeline_stream_on_the_device(rt):
    """All contexts of a process on one device share ONE stream; rtgl_set_stream must refuse a binding that would let two
    path-tracing pipelines run concurrently, accept the same stream for all of them, and step aside when the caller
    takes the ordering over (RTGL_AMD_ALLOW_CONCURRENT_PIPELINES=1)."""
    import os
    dev = torch.device("cuda", 0)
    a = rt.host.Context(64, 64, device=0)
    b = rt.host.Context(64, 64, device=0)
    side = torch.cuda.Stream(device=dev)
    with pytest.raises(rt.host.RtglError, match="different stream"):
        a.set_stream(side.cuda_stream)                 # b still renders on the library's stream
    os.environ["RTGL_AMD_ALLOW_CONCURRENT_PIPELINES"] = "1"
    try:
        a.set_stream(side.cuda_stream)                 # the caller has taken the ordering over
    finally:
        del os.environ["RTGL_AMD_ALLOW_CONCURRENT_PIPELINES"]
    b.set_stream(side.cuda_stream)                     # the SAME stream for both: fine without the override
    a.set_stream(0); b.set_stream(0)
    a.close(); b.close()
