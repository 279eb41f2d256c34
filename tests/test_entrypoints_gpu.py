"""GPU: reference-compatible entry points -- scenario pkl round trip against the reference's own fixture
format, the vectorised evaluation loop against a sequential evaluation through the single-env facade, and a
short end-to-end train.py run."""
import os
import pickle
import subprocess
import sys

import numpy as np
import pytest
import torch

from tests import helpers as H

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tests_from_golden(nag, cases=3):
    z = H.load_npz("env_fixtures.npz")
    t = {"maps": [], "agents": [], "goals": []}
    for c in range(cases):
        pre = "fix%d_c%d_" % (nag, c)
        t["maps"].append(z[pre + "map"].astype(np.float32))
        t["agents"].append(z[pre + "agents"].astype(np.int64))
        t["goals"].append(z[pre + "goals"].astype(np.int64))
    return t


def test_pkl_round_trip(tmp_path):
    from mapf_rl_amd import evaluate as EV

    t = _tests_from_golden(16)
    p = str(tmp_path / "test16_40.pkl")
    EV.save_tests(p, t["maps"], t["agents"], t["goals"])
    back = EV.load_tests(p)
    assert all(np.array_equal(a, b) for a, b in zip(back["maps"], t["maps"]))
    assert back["agents"][0].dtype == np.int64 and back["agents"][0].shape == (16, 2)
    plain = pickle.load(open(p, "rb"))  # what the reference's test.py does (test.py:89-90)
    assert set(plain.keys()) == {"maps", "agents", "goals"}
    made = EV.create_test(5, 12, test_num=7, density=0.2, seed=3, path=str(tmp_path / "t.pkl"))
    assert len(made["maps"]) == 7 and made["maps"][0].shape == (12, 12) and made["agents"][0].shape == (5, 2)
    bad = str(tmp_path / "bad.pkl")
    pickle.dump({"maps": os.system}, open(bad, "wb"))
    with pytest.raises(pickle.UnpicklingError):
        EV.load_tests(bad)


@pytest.mark.parametrize("nag", [16, 64])
def test_vectorised_evaluation_equals_sequential(nag):
    """test_model's loop (reference test.py:105-143) vectorised over cases == the same loop run case by case
    through the reference-compatible single-env facade with the same network; 64 = the reference's largest fixture
    (test64_40_0.3.pkl), whose recurrence runs through the wide kernels."""
    import mapf_rl_amd as M
    from mapf_rl_amd import evaluate as EV
    from mapf_rl_amd.model import Network

    torch.manual_seed(3)
    net = Network().cuda().eval()
    tests = _tests_from_golden(nag)
    f_rate, mean_steps, steps, ok = EV.evaluate(net, tests, max_steps=12, num_cases=3)
    for i in range(3):
        env = M.Environment()
        env.load(tests["maps"][i], tests["agents"][i], tests["goals"][i])
        done = False
        net.reset()
        while not done and env.steps < 12:
            obs, pos = env.observe()
            actions, _, _, _ = net.step(torch.from_numpy(obs.astype(np.float32)), torch.from_numpy(pos.astype(np.float32)))
            _, _, done, _ = env.step(actions)
        assert env.steps == steps[i]
        assert bool(np.array_equal(env.agents_pos, env.goals_pos)) == bool(ok[i])
    assert 0.0 <= f_rate <= 1.0 and mean_steps <= 12


def test_train_entry_point_runs(tmp_path):
    """`python train.py` for a few seconds on a tiny configuration: collects, starts training, checkpoints."""
    env = dict(os.environ, PYTHONPATH=ROOT)
    cmd = [sys.executable, os.path.join(ROOT, "train.py"), "--envs", "64", "--agents", "2", "--map", "10", "--capacity", "256",
           "--learning-starts", "2000", "--batch-size", "16", "--max-updates", "6", "--interval", "2"]
    out = subprocess.run(cmd, cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "start training" in out.stdout
    assert os.path.exists(str(tmp_path / "models" / "6.pth"))
    sd = torch.load(str(tmp_path / "models" / "6.pth"), map_location="cpu")
    assert "obs_encoder.0.weight" in sd and "comm.self_attn.W_Q.weight" in sd and len(sd) == 35


def test_train_default_is_the_curriculum(tmp_path):
    """`python train.py` without a level = the reference's adaptive schedule from config.init_set (worker.py:362)."""
    env = dict(os.environ, PYTHONPATH=ROOT)
    cmd = [sys.executable, os.path.join(ROOT, "train.py"), "--envs", "64", "--learning-starts", "1500", "--batch-size", "16",
           "--max-updates", "4", "--interval", "0.05"]
    out = subprocess.run(cmd, cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "start training" in out.stdout and "(1, 10): " in out.stdout
    assert os.path.exists(str(tmp_path / "models" / "4.pth"))


def test_train_two_ranks_stop_together(tmp_path):
    """Two ranks (gloo, sharing GPU 0) with different scenarios per rank: start, statistics and the --minutes stop are
    collective decisions, so both ranks run the same number of all-reducing updates and exit cleanly."""
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(2):
        env = dict(os.environ, PYTHONPATH=ROOT, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), MAPF_TRAIN_SHARE_GPU="1")
        cmd = [sys.executable, os.path.join(ROOT, "train.py"), "--envs", "48" if r == 0 else "32", "--agents", "2", "--map", "10",
               "--capacity", "256", "--learning-starts", "1500", "--batch-size", "16", "--minutes", "0.2", "--interval", "2",
               "--dist-backend", "gloo"]
        procs.append(subprocess.Popen(cmd, cwd=str(tmp_path), env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    assert [p.returncode for p in procs] == [0, 0], (outs[0][1][-1500:], outs[1][1][-1500:])
    assert "start training" in outs[0][0] and "number of updates" in outs[0][0] and "(2, 10): " in outs[0][0]
    assert "number of updates" not in outs[1][0] and "(2, 10): " not in outs[1][0]  # only rank 0 prints statistics
    assert len(os.listdir(str(tmp_path / "models"))) >= 1


def _bench_env():
    env = dict(os.environ, PYTHONPATH=ROOT, MAPF_BENCH_SHARE_GPU="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MAPF_BENCH_FAULT"):
        env.pop(k, None)
    return env


SMALL_BENCH = ["--steps", "5", "--warmup", "2", "--envs", "256", "--dist-backend", "gloo", "--no-cpu-baseline", "--dqn-updates", "6",
               "--dqn-actor-iters", "4", "--train-iters", "8", "--curriculum-envs", "64", "--curriculum-iters", "20", "--ref-shape-updates", "10",
               "--ref-shape-warmup", "12"]


def test_bench_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` without a launcher starts the two ranks itself (fresh child processes, before it touches the
    GPU) and forwards rank 0's JSON lines; here both ranks share GPU 0 over gloo (MAPF_BENCH_SHARE_GPU=1, the single-GPU rehearsal
    of the driver's N-GPU command).  Several ranks: the headline line first (`partial`), the complete line last."""
    import json

    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + SMALL_BENCH
    out = subprocess.run(cmd, cwd=str(tmp_path), env=_bench_env(), capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 2
    h, r = json.loads(lines[0]), json.loads(lines[-1])
    assert h["partial"] is True and "partial" not in r and "dqn_error" not in r
    assert h["value"] == r["value"] and h["roofline"]["frac"] == r["roofline"]["frac"] and "learner_updates_per_sec" not in h
    assert r["n_gpus"] == 2 and "dist.get_world_size()=2" in r["config"]["parallelism"] and r["steps"] == 5
    assert r["value"] > 0 and r["learner_updates_per_sec"] > 0 and "flat-bucket" in r["learner_config"] and r["learner_path"] == "fused"
    # the timed region is stretched to >= ~12 ms by repeating the K-step tape; the pruning spread over ranks is reported
    assert r["timed_repeats"] >= 1 and r["timed_region_ms"] >= 10.0
    assert abs(r["value"] - 2 * 256 * 5 * r["timed_repeats"] / (r["timed_region_ms"] * 1e-3)) <= 1e-6 * r["value"]
    assert 0 < r["learner_reachable_fraction_min"] <= r["learner_reachable_fraction"] * (1 + 1e-9) + 1e-12
    assert r["learner_reachable_fraction_min"] <= r["learner_reachable_fraction_max"] <= 1
    rf = r["roofline"]
    assert rf["envs_out_of_cache"] >= 4 * 256 and rf["working_set_out_of_cache_bytes"] >= 2 * (256 << 20)
    assert 0 < rf["frac_out_of_cache"] < 1 and 0 < rf["frac"] < 1.2
    # who took part, on what, and what the exchange cost per update (per-N visibility)
    mr = r["multi_rank"]
    assert mr["backend"] == "gloo" and mr["world_size"] == 2 and [x["rank"] for x in mr["ranks_seen"]] == [0, 1]
    assert mr["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and "rccl_version" in mr
    assert r["learner_exchange_pieces_per_update"] == 2 and r["learner_exchange_begin_ms"] >= 0 and r["learner_exchange_finish_ms"] > 0
    assert r["learner_ref_shape_exchange_pieces_per_update"] == 2 and r["learner_ref_shape_exchange_finish_ms"] > 0  # (graph mode: behind the backward stage)
    # the interleaved loop costs what its two halves cost (round-4 review: 410 ms beside a 14 ms update and a 4 ms actor iteration --
    # stalls inside PyTorch's gloo path for device tensors, profiles/r05_two_rank_probe.md; the exchange is host-staged through one
    # persistent pinned buffer now, learner.FlatGradBucket._host_staged)
    assert r["train_loop_ms_per_iter"] <= 1.5 * (r["learner_ms_per_update"] + r["actor_loop_ms_per_iter"]), (
        r["train_loop_ms_per_iter"], r["learner_ms_per_update"], r["actor_loop_ms_per_iter"])
    assert r["train_loop_tape_policy_ms_per_iter"] <= 1.5 * (r["learner_ms_per_update"] + r["actor_loop_tape_policy_ms_per_iter"])
    # the reference's own training shape: graph-replayed update on the curriculum actors' replay, the captured actor iteration
    assert r["learner_ref_shape_path"] == "fused" and r["learner_ref_shape_ms_per_update"] > 0 and r["learner_ref_shape_graph_replays"] == 10
    assert r["curriculum_actor_iter_ms"] > 0 and r["curriculum_actor_graph_replays"] == 20 and r["train_loop_ref_shape_ms_per_iter"] > 0


def test_bench_rank_fault_ends_the_job_with_the_headline_out(tmp_path):
    """A rank that raises inside a secondary leg (here rank 1, at the start of the learner leg, while rank 0 walks into the leg's
    first collective) leaves the job at once: non-zero exit within seconds, never a hang until the driver's deadline -- and the
    headline line is already on stdout."""
    import json
    import time

    env = dict(_bench_env(), MAPF_BENCH_FAULT="1:learner")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--no-out-of-cache"] + SMALL_BENCH
    t0 = time.time()
    out = subprocess.run(cmd, cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=900)
    dt = time.time() - t0
    assert out.returncode == 1
    assert "injected fault at stage 'learner' on rank 1" in out.stderr and "leaving the job (exit 13)" in out.stderr
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    h = json.loads(lines[0])
    assert h["partial"] is True and h["metric"] == "env_steps_per_sec" and h["value"] > 0 and h["roofline"]["frac"] > 0 and h["n_gpus"] == 2
    # (bounded by the legs in front of the fault, not by a timeout: the whole two-rank command takes ~1-2 minutes when it succeeds)
    assert dt < 240, dt
    # (rank 0 either gets ended by the launcher or -- over gloo -- sees its peer's connection drop inside the all-reduce and leaves the
    # same way by itself)
    assert "rank exit codes" in out.stderr and ("ending the other ranks" in out.stderr or "[13, 13]" in out.stderr), out.stderr[-1500:]


def test_one_rank_bench_prints_one_line(tmp_path):
    """The driver's N=1 command shape: exactly ONE JSON line, with the reference-training-shape keys and the moving-policy train loop."""
    import json

    env = _bench_env()
    env.pop("MAPF_BENCH_SHARE_GPU")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + [a for a in SMALL_BENCH if a not in ("--dist-backend", "gloo")]
    out = subprocess.run(cmd, cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    r = json.loads(lines[0])
    assert "partial" not in r and "multi_rank" not in r and "dqn_error" not in r and r["n_gpus"] == 1
    for k in ("learner_ref_shape_ms_per_update", "learner_ref_shape_graph_captures", "curriculum_actor_iter_ms", "train_loop_tape_policy_ms_per_iter",
              "train_loop_tape_policy_updates_per_sec", "train_loop_ref_shape_updates_per_sec", "learner_path", "pipeline_env_steps_per_sec"):
        assert k in r, k
    assert r["learner_path"] == "fused" and r["learner_ref_shape_path"] == "fused"
