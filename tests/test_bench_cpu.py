"""CPU: bench.py's rank launcher refuses to start more ranks than there are HIP devices -- before anything touches a GPU --
and exits non-zero (the driver must not record N = 1 numbers under an N-GPU command)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_more_ranks_than_devices_is_an_error():
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MAPF_BENCH_SHARE_GPU"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64"], capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 2 and "HIP device" in out.stderr and out.stdout.strip() == ""
