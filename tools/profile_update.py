#!/usr/bin/env python3
"""Runs a few learner updates / actor iterations at the config-2 shape for rocprofv3 (kernel breakdown)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import mapf_rl_amd as M
from mapf_rl_amd.actor import VecActor
from mapf_rl_amd.learner import Learner
from mapf_rl_amd.replay import GlobalBuffer
N = int(os.environ.get("NAGENTS", 40))
dev = torch.device("cuda")
buf = GlobalBuffer(64, max_agents=N, device=dev)
g2 = torch.Generator(device=dev); g2.manual_seed(5)
RD, CW, S = buf.row_dwords, (N + 31) // 32, 96
for k in range(64):
    td = torch.zeros(256, dtype=torch.float64, device=dev); td[:S] = torch.rand(S, generator=g2, device=dev, dtype=torch.float64) + 0.05
    buf.add_episode_device(N, S, k % 2, torch.randint(-2**31, 2**31 - 1, (S + 1, RD), generator=g2, device=dev, dtype=torch.int32) &
                           torch.randint(-2**31, 2**31 - 1, (S + 1, RD), generator=g2, device=dev, dtype=torch.int32),
                           torch.randint(0, 2**20, (S + 1, N, CW), generator=g2, device=dev, dtype=torch.int32),
                           torch.randint(0, 5, (S,), generator=g2, device=dev, dtype=torch.uint8),
                           (torch.rand(S, generator=g2, device=dev) - 0.5).half(), (torch.randn((S, 256), generator=g2, device=dev) * 0.3).half(), td)
lr = Learner(buf, device=dev, batch_size=192)
for _ in range(int(os.environ.get("TUPD", 4))):
    lr.update()
torch.cuda.synchronize()
if os.environ.get("TACT"):
    E = 4096
    maps, agents, goals, _ = M.generate_scenarios(E, 32, N, 0.3, seed=1)
    env = M.VecEnvironment(E, 32, N); env.load(maps, agents, goals)
    actor = VecActor(env, lr.model, None, seed=0, density=0.3)
    for _ in range(3):
        actor.step()
    torch.cuda.synchronize()
