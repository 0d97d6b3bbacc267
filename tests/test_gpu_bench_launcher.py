"""`python bench.py --gpus 2 --steps 8` as typed, on the one-GPU box: the launcher starts two ranks; the TEST-ONLY mapping
RTGL_BENCH_SHARED_DEVICE=1 puts both on device 0 (RCCL refuses two ranks on one device, so the tile buffers travel through the host with
gloo there).  One JSON line with n_gpus 2 must come out, and the assembled image path must have run on both ranks."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_with_two_ranks_runs_as_typed():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["RTGL_BENCH_SHARED_DEVICE"] = "1"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "8", "--warmup", "2"], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["steps"] == 8 and d["warmup"] == 2
    assert d["collective_backend"] == "gloo" and "TEST MAPPING" in d["config"]["parallelism"]
    assert d["value"] > 0 and d["scaling"] == "strong"
    assert d["counters_per_frame"]["paths"] == 1920 * 1080          # both ranks' strips together are the whole frame
