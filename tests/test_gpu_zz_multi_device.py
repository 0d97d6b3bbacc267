"""rtgl_create_multi on REAL distinct device ordinals: peer access, the cross-device event waits, the 2-D peer copies of the
gather, one submit thread per device, and the ordering of a frame's writes behind the previous gather's reads (the parts' streams wait
for `gather_done`).  Every other test of the multi-device context runs all parts on device 0 (tests/test_gpu_fullsize.py); the one-GPU
test box skips these -- they are here for multi-GPU nodes, and sort last so that nothing else hides behind them."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def visible_devices():
    n = ctypes.c_int(0)
    try:
        hip = ctypes.CDLL("libamdhip64.so")
        return n.value if hip.hipGetDeviceCount(ctypes.byref(n)) == 0 else 0
    except OSError:
        return 0


@pytest.mark.parametrize("devices", [[0, 1], [1, 0, 1], [0, 1, 2, 3]])
def test_multi_device_context_on_distinct_devices_matches_single_context(devices, rt):
    if visible_devices() <= max(devices):
        pytest.skip(f"needs {max(devices) + 1} visible devices")
    sc = rt.scenes
    W, H = 328, 204
    scene = sc.scene_mesh(30, 10, env_size=32)
    base = sc.params_c2()
    g = sc.GlibcRand(0)
    plist = [base.replace(frames=f, random=g.rand()) for f in range(1, 6)]

    def run(**kw):
        ctx = rt.host.Context(W, H, **kw)
        ctx.upload_scene(scene)
        imgs = []
        for p in plist:
            ctx.render(p)
            imgs.append(ctx.read_image().copy())           # a gather after every frame: the next frame's writes must wait for its copies
        ctx.close()
        return imgs

    ref = run()
    got = run(devices=devices, strip_rows=8)
    for f, (a, b) in enumerate(zip(ref, got)):
        assert (a.view(np.uint32) == b.view(np.uint32)).all(), f"frame {f + 1} differs"
