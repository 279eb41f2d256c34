#!/usr/bin/env python3
"""Runs K launches of mapf_step on the BASELINE config-2 workload and nothing else (for rocprofv3 --pmc runs)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mapf_rl_amd as M  # noqa: E402

E, L, N = int(os.environ.get("TE", 4096)), int(os.environ.get("TL", 32)), int(os.environ.get("TN", 40))
K = int(os.environ.get("TK", 20))
maps, agents, goals, _ = M.generate_scenarios(E, L, N, 0.3, seed=1)
env = M.VecEnvironment(E, L, N)
env.load(maps, agents, goals)
tape = torch.randint(0, 5, (K, E, N), dtype=torch.int8, device="cuda")
for k in range(K):
    env.step(tape[k])
torch.cuda.synchronize()
env.check_status()
