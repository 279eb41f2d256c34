#!/usr/bin/env python3
"""Is the config-2 update held back by the host?  Wall and host-enqueue time per update (bench-like replay content: the tape policy's
episodes), eager against graph replay (FusedUpdate.GRAPH_MAX_AGENTS raised).  Run on the GPU."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mapf_rl_amd as M  # noqa: E402
from mapf_rl_amd.actor import VecActor  # noqa: E402
from mapf_rl_amd.learner import Learner  # noqa: E402
from mapf_rl_amd.replay import GlobalBuffer  # noqa: E402
from mapf_rl_amd.update import FusedUpdate  # noqa: E402

E, N, L = 2048, 40, 32
dev = torch.device("cuda")
FusedUpdate.GRAPH_MAX_AGENTS = int(os.environ.get("GMAX", "16"))
env = M.VecEnvironment(E, L, N, device=dev)
maps, agents, goals, _ = M.generate_scenarios(E, L, N, 0.3, seed=1)
env.load(maps, agents, goals)
buf = GlobalBuffer(1 << (2 * E - 1).bit_length(), max_agents=N, device=dev, init_set=(N, L), fixed_level=True)
learner = Learner(buf, device=dev, batch_size=192)
actor = VecActor(env, learner.model, buf, seed=0, density=0.3, weights_period=400)
g = torch.Generator(device=dev).manual_seed(0)
for _ in range(300):
    actor.step(actions_override=torch.randint(0, 5, (E, N), device=dev, generator=g))
torch.cuda.synchronize()
for _ in range(40):
    learner.update()
torch.cuda.synchronize()
n = 60
t0 = time.perf_counter()
for _ in range(n):
    learner.update()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
f = learner._fused
print("GRAPH_MAX_AGENTS=%d graph_mode=%s: update %.2f ms wall, host enqueue %.2f ms; captures %d" % (FusedUpdate.GRAPH_MAX_AGENTS, f.graph_mode(), (t2 - t0) / n * 1e3,
                                                                                           (t1 - t0) / n * 1e3, getattr(f, "graph_captures", 0)))
