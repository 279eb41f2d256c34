#!/usr/bin/env python3
"""Where does the fixed-level actor iteration's wall time go when few observations change (greedy actions of a random-init network)?
Wall per iteration, host enqueue per iteration (no sync inside the loop) and a cProfile of the host side.  Run on the GPU."""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mapf_rl_amd.actor import VecActor  # noqa: E402
from mapf_rl_amd.environment import VecEnvironment, generate_scenarios  # noqa: E402
from mapf_rl_amd.learner import Learner  # noqa: E402
from mapf_rl_amd.replay import GlobalBuffer  # noqa: E402

E, N, L = int(os.environ.get("NENVS", 4096)), int(os.environ.get("NAGENTS", 40)), int(os.environ.get("MAPLEN", 32))
dev = torch.device("cuda")
env = VecEnvironment(E, L, N, device=dev)
maps, agents, goals, _ = generate_scenarios(E, L, N, 0.3, seed=1)
env.load(maps, agents, goals)
buf = GlobalBuffer(1 << (2 * E - 1).bit_length(), max_agents=max(N, 6), device=dev, init_set=(N, L), fixed_level=True)
learner = Learner(buf, device=dev, batch_size=192)
actor = VecActor(env, learner.model, buf, seed=0, density=0.3, weights_period=400)
for _ in range(80):
    actor.step()
torch.cuda.synchronize()
n = 200
t0 = time.perf_counter()
for _ in range(n):
    actor.step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("%d envs x %d agents: wall %.3f ms per iteration, host enqueue %.3f ms per iteration" % (E, N, (t2 - t0) / n * 1e3, (t1 - t0) / n * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(n):
    actor.step()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
