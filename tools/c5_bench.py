#!/usr/bin/env python3
"""BASELINE config 5 on one GPU (64x64 grid, 128 agents, 2048 environments; prioritized replay, optional double-DQN): env-step
kernel, full actor loop and learner update times.  The recurrence of 128-agent environments runs through
csrc/mapf_recur_wide*.hip.  Usage: c5_bench.py [agents] [map] [envs] [--double-q]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mapf_rl_amd as M  # noqa: E402
from mapf_rl_amd.actor import VecActor  # noqa: E402
from mapf_rl_amd.learner import Learner  # noqa: E402
from mapf_rl_amd.replay import GlobalBuffer  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
N = int(args[0]) if len(args) > 0 else 128
L = int(args[1]) if len(args) > 1 else 64
E = int(args[2]) if len(args) > 2 else 2048
dev = torch.device("cuda", 0)
torch.manual_seed(0)
env = M.VecEnvironment(E, L, N, device=dev)
env.reset_envs(None, 0.3, seed=1)
env.check_status()
buf = GlobalBuffer(64, max_agents=N, device=dev)
g2 = torch.Generator(device=dev)
g2.manual_seed(5)
RD, CW, S = buf.row_dwords, (N + 31) // 32, 96
for k in range(64):
    td = torch.zeros(256, dtype=torch.float64, device=dev)
    td[:S] = torch.rand(S, generator=g2, device=dev, dtype=torch.float64) + 0.05
    buf.add_episode_device(
        N, S, k % 2, torch.randint(-2**31, 2**31 - 1, (S + 1, RD), generator=g2, device=dev, dtype=torch.int32) &
        torch.randint(-2**31, 2**31 - 1, (S + 1, RD), generator=g2, device=dev, dtype=torch.int32),
        torch.randint(0, 2**20, (S + 1, N, CW), generator=g2, device=dev, dtype=torch.int32),
        torch.randint(0, 5, (S,), generator=g2, device=dev, dtype=torch.uint8),
        (torch.rand(S, generator=g2, device=dev) - 0.5).half(), (torch.randn((S, 256), generator=g2, device=dev) * 0.3).half(), td)
learner = Learner(buf, device=dev, batch_size=192, double_q="--double-q" in sys.argv)
for _ in range(2):
    learner.update()
torch.cuda.synchronize()
t0 = time.perf_counter()
K = 4
for _ in range(K):
    learner.update()
torch.cuda.synchronize()
dt_upd = (time.perf_counter() - t0) / K
actor = VecActor(env, learner.model, None, seed=0, density=0.3)
actor.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(K):
    actor.step()
torch.cuda.synchronize()
dt_act = (time.perf_counter() - t0) / K
env.check_status()
print("C5 shape %dx%d, %d agents, %d envs, double_q=%s: learner %.1f ms/update (%.2f updates/s, B=192 x T=18 x A=%d), actor loop %.2f ms/iter "
      "(%.3g env-steps/s), peak memory %.1f GB" % (L, L, N, E, learner.double_q, dt_upd * 1e3, 1 / dt_upd, N, dt_act * 1e3, E / dt_act,
                                                   torch.cuda.max_memory_allocated() / 1e9), flush=True)
