#!/usr/bin/env python3
"""Runs a few actor-loop iterations (Network.step_batch + mapf_step + recording) at the config-2 shape (NAGENTS / NENVS / MAPLEN: another) for
rocprofv3 (kernel breakdown of the end-to-end env-steps/s)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mapf_rl_amd as M  # noqa: E402
from mapf_rl_amd.actor import VecActor  # noqa: E402
from mapf_rl_amd.model import Network  # noqa: E402

N, E, L = int(os.environ.get("NAGENTS", 40)), int(os.environ.get("NENVS", 4096)), int(os.environ.get("MAPLEN", 32))
torch.manual_seed(0)
model = Network().cuda()
maps, agents, goals, _ = M.generate_scenarios(E, L, N, 0.3, seed=1)
env = M.VecEnvironment(E, L, N)
env.load(maps, agents, goals)
actor = VecActor(env, model, None, seed=0, density=0.3)
import time
n = int(os.environ.get("TACT", 6))
actor.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    actor.step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("actor loop: %.3f ms per iteration (host enqueue %.3f ms per iteration), %d iterations" % ((t2 - t0) / n * 1e3, (t1 - t0) / n * 1e3, n))
