"""CPU: bench.py's own rank launcher (`python bench.py --gpus N` without torch.distributed.run) cannot hang and cannot lose the
headline: one rank exiting non-zero -- or the launcher's deadline -- ends every rank within seconds, and whatever rank 0 had printed
until then is forwarded.  The ranks here are a stub script (no GPU); the real ranks' fail-fast path (a rank that catches an exception
in a secondary leg exits 13 at once) is tests/test_entrypoints_gpu.py::test_bench_rank_fault_ends_the_job_with_the_headline_out."""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

STUB = r'''
import os, sys, time
rank, mode = int(os.environ["RANK"]), sys.argv[1]
assert os.environ["WORLD_SIZE"] == sys.argv[2] and os.environ["MASTER_ADDR"] == "127.0.0.1" and int(os.environ["MASTER_PORT"]) > 0
assert os.environ["LOCAL_RANK"] == os.environ["RANK"] and os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
if rank == 0:
    print('{"metric": "env_steps_per_sec", "partial": true}', flush=True)
if mode == "ok":
    time.sleep(0.3)
    if rank == 0:
        print('{"metric": "env_steps_per_sec", "full": true}', flush=True)
    sys.exit(0)
if mode == "fault" and rank == int(sys.argv[3]):
    time.sleep(1.0)
    os._exit(13)
time.sleep(120)  # the other ranks "wait in a collective"
'''


def _launch(tmp_path, n, mode, *extra, deadline=None):
    stub = tmp_path / "stub_rank.py"
    stub.write_text(STUB)
    code = ("import sys; sys.path.insert(0, %r); import bench; sys.exit(bench.launch_ranks(%d, [%r, %r] + %r, script=%r, deadline_s=%r, grace_s=0.5))"
            % (ROOT, n, mode, str(n), [str(e) for e in extra], str(stub), deadline))
    env = dict(os.environ, MAPF_BENCH_SHARE_GPU="1")  # (no HIP device here: the launcher must not insist on N of them)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    t0 = time.time()
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    return out, time.time() - t0


def test_all_ranks_fine_forwards_rank0_lines_in_order(tmp_path):
    out, dt = _launch(tmp_path, 8, "ok")
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 2 and '"partial": true' in lines[0] and '"full": true' in lines[1]


def test_one_failing_rank_ends_the_job_and_keeps_the_headline(tmp_path):
    out, dt = _launch(tmp_path, 8, "fault", 5)
    assert out.returncode == 1
    assert dt < 30, dt  # (the other seven ranks would sleep for 120 s)
    assert '"partial": true' in out.stdout
    assert "rank(s) [5] exited non-zero" in out.stderr and "rank exit codes" in out.stderr


def test_launcher_deadline_ends_a_hung_job(tmp_path):
    out, dt = _launch(tmp_path, 3, "hang", deadline=3.0)
    assert out.returncode == 1 and dt < 30, dt
    assert '"partial": true' in out.stdout and "launcher deadline reached" in out.stderr


def test_launcher_deadline_is_inside_the_drivers():
    sys.path.insert(0, ROOT)
    import bench

    assert bench.LAUNCH_DEADLINE_S < 600 and bench.WATCHDOG_DEFAULT_S < bench.LAUNCH_DEADLINE_S
