#!/usr/bin/env python3
"""bench.py -- env steps/s of the vectorised MAPF environment hot path on MI355X.

Workload (BASELINE.json configs[1]): 32x32 grid, 40 agents, obstacle density 0.3, 4096 lock-step
environments per GPU.  One "step" = one pass of the fused step+observe kernel (mapf_step) over the
4096 resident environments, replaying a pre-recorded action tape (80 % heuristic-following / 20 %
uniform); inputs are resident in HBM before the timed region.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Multi-GPU: environments are independent, so rank g owns its own 4096 envs (weak scaling); there is no
data-path collective, only the timing barrier / max-over-ranks.

Prints ONE JSON line on rank 0 (see the driver contract), including `roofline` for the dominant kernel
(env_step_kernel, HBM-bound) and `cpu_baseline` (the CPU oracle timed on the host cores of this box).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def heuristic_actions(obs, gen, p_follow=0.8):
    """80 % follow a navi flag of the own cell (obs[:, :, 2:6, 4, 4]), 20 % uniform."""
    E, N = obs.shape[:2]
    flags = obs[:, :, 2:6, 4, 4] != 0
    score = torch.rand((E, N, 4), device=obs.device, generator=gen) * flags
    follow = torch.where(flags.any(-1), 1 + score.argmax(-1), torch.zeros((), dtype=torch.long, device=obs.device))
    uni = torch.randint(0, 5, (E, N), device=obs.device, generator=gen)
    pick = torch.rand((E, N), device=obs.device, generator=gen) < p_follow
    return torch.where(pick, follow, uni).to(torch.int8).contiguous()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--envs", type=int, default=4096, help="environments per GPU")
    ap.add_argument("--map", type=int, default=32)
    ap.add_argument("--agents", type=int, default=40)
    ap.add_argument("--density", type=float, default=0.3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU-baseline wall time")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    assert torch.cuda.is_available(), "bench.py needs a HIP device"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    import mapf_rl_amd as M

    E, L, N, K, W = args.envs, args.map, args.agents, args.steps, args.warmup
    T = K + W
    t0 = time.time()
    maps, agents, goals, redraws = M.generate_scenarios(E, L, N, args.density, seed=1000 + rank)
    env = M.VecEnvironment(E, L, N, device=dev)
    env.load(maps, agents, goals)
    torch.cuda.synchronize()
    log("[rank %d] scenarios + navi ready in %.2fs (redraws %d)" % (rank, time.time() - t0, redraws))

    # ---- record the action tape by running the policy once (untimed) ----
    gen = torch.Generator(device=dev)
    gen.manual_seed(77 + rank)
    tape = torch.empty((T, E, N), dtype=torch.int8, device=dev)
    obs, pos = env.observe()
    for t in range(T):
        tape[t] = heuristic_actions(obs, gen)
        obs, pos, rew, done, rc = env.step(tape[t])
    env.check_status()
    final_pos_first_pass = pos.clone()
    agents_dev = torch.from_numpy(agents).to(dev)

    # ---- timed replay ----
    env.set_agents(agents_dev)
    for t in range(W):
        env.step(tape[t])
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(K)]
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t_start = time.perf_counter()
    for k in range(K):
        ev[k][0].record()
        env.step(tape[W + k])
        ev[k][1].record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t_start
    env.check_status()
    assert torch.equal(env.pos, final_pos_first_pass), "replay diverged from the recording pass"

    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    kern_ms = np.array([a.elapsed_time(b) for a, b in ev])
    kern_avg_s = float(kern_ms.mean()) * 1e-3
    alg_bytes_per_env = L * L + 821 * N + 1  # fused step+observe, one byte per flag/cell
    alg_bytes = alg_bytes_per_env * E
    achieved = alg_bytes / kern_avg_s / 1e9

    result = {
        "metric": "env_steps_per_sec",
        "value": world * E * K / elapsed,
        "unit": "env-steps/s",
        "n_gpus": world,
        "steps": K,
        "warmup": W,
        "ms_per_step": elapsed / K * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u8",
        "data": "synthetic",
        "config": {"workload": "mapf env step+observe, %dx%d grid, %d agents, rho=%.2f, %d envs/GPU" % (L, L, N, args.density, E),
                   "map": L, "agents": N, "envs_per_gpu": E, "obs_radius": 4, "parallelism": "env-sharded x%d" % world},
        "roofline": {"bound": "hbm", "kernel": "env_step_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                     "alg_bytes_per_launch": alg_bytes, "kernel_avg_us": kern_avg_s * 1e6,
                     "kernel_min_us": float(kern_ms.min()) * 1e3},
    }
    tr = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tr):
        try:
            result["roofline"]["traffic"] = json.load(open(tr)).get("env_step_kernel_bytes_per_launch")
        except Exception:
            pass

    # ---- CPU baseline: the oracle (C restatement with the reference's sequential semantics) ----
    if rank == 0 and not args.no_cpu_baseline:
        from oracle import oracle

        S = min(E, 256)
        nthreads = oracle.max_threads()
        tape_h = tape[:, :S].cpu().numpy()
        nv = oracle.navi_batch(maps[:S], goals[:S], nthreads)
        # trajectories must be identical to the GPU's before any number is reported
        chk = oracle.rollout(maps[:S], agents[:S], goals[:S], nv, tape_h, want_pos=False, want_rclass=False,
                             want_done=False, want_hash=True, nthreads=nthreads)
        assert chk["status"] == 0
        assert np.array_equal(chk["final_agents"], final_pos_first_pass[:S].cpu().numpy()), "CPU/GPU trajectories differ"
        t1 = time.perf_counter()
        reps = 0
        while True:
            oracle.rollout(maps[:S], agents[:S], goals[:S], nv, tape_h, want_pos=False, want_rclass=False,
                           want_done=False, want_hash=True, nthreads=nthreads)
            reps += 1
            dt = time.perf_counter() - t1
            if dt >= args.cpu_seconds or reps >= 10000:
                break
        result["cpu_baseline"] = {
            "value": S * T * reps / dt, "unit": "env-steps/s", "cores": nthreads, "kind": "port",
            "sample": "first %d envs x %d tape steps x %d repeats (%.1f s), step+observe, OpenMP one env per thread; "
                      "trajectories verified identical to the GPU run" % (S, T, reps, dt),
        }
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
