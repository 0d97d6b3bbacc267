"""`python bench.py --gpus N` typed without a launcher must start the N ranks itself: the command it builds, and that
the parent relays the child's return code without importing torch or touching a GPU."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_relaunch_command_is_one_rank_per_gpu_on_loopback():
    import bench
    cmd = bench.relaunch_command(["--gpus", "4", "--steps", "8", "--warmup", "2"], 4, port=29999)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29999"
    assert cmd[-7] == os.path.join(ROOT, "bench.py") and cmd[-6:] == ["--gpus", "4", "--steps", "8", "--warmup", "2"]


def test_parent_relays_the_return_code_of_the_rank_job_without_a_gpu():
    """No GPU here: every rank exits with bench.py's "needs a HIP device" error; the parent must hand that failure on (and must not
    itself have needed torch.cuda).  A WORLD_SIZE in the environment means "I am a rank": no relaunch."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "8", "--warmup", "1"], env=env, capture_output=True, text=True, timeout=300)
    import torch
    if torch.cuda.is_available():
        return                       # (on a GPU box the GPU test covers the launcher end to end)
    assert out.returncode != 0
    assert "needs a HIP device" in (out.stderr + out.stdout)
    env["WORLD_SIZE"] = "1"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "8"], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and "does not match WORLD_SIZE" in (out.stderr + out.stdout)
