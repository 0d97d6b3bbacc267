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
    # detach again before the tensor goes away
    ctx.bind_device_image(0)
    ctx.set_stream(0)
    ctx.close()
