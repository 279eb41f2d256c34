#!/usr/bin/env python3
"""Host-enqueue time vs wall time of one actor iteration and one learner update at the curriculum's few-agent
levels (the reference's own training regime, config.py:5-6,49-52).  When the two are equal the loop is bound by
the host's launch rate, not by the GPU.  Usage: small_shape_times.py [envs]   (MAPF_GRAPHS=0/1 is honoured)"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import config  # noqa: E402
import mapf_rl_amd as M  # noqa: E402
from mapf_rl_amd.actor import VecActor  # noqa: E402
from mapf_rl_amd.learner import Learner  # noqa: E402
from mapf_rl_amd.replay import GlobalBuffer  # noqa: E402

E = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = torch.device("cuda")


def timed(fn, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3


print("| level (agents, map) | envs | actor host ms | actor wall ms | update host ms | update wall ms | both wall ms |\n|---|---|---|---|---|---|---|")
for n_agents, map_len in ((1, 10), (3, 15), (6, 20), (6, 40)):
    torch.manual_seed(0)
    env = M.VecEnvironment(E, map_len, n_agents, config.obs_radius, config.reward_fn, device=dev)
    maps, agents, goals, _ = M.generate_scenarios(E, map_len, n_agents, -1.0, seed=3)
    env.load(maps, agents, goals)
    buf = GlobalBuffer(4096, max_agents=config.max_num_agetns, device=dev, init_set=(n_agents, map_len), fixed_level=True)
    lr = Learner(buf, device=dev, batch_size=config.batch_size)
    actor = VecActor(env, lr.model, buf, seed=1)
    for _ in range(300):
        actor.step()
    assert len(buf) > 192 * 4, len(buf)
    for _ in range(5):
        lr.update()
    ah, aw = timed(actor.step, 100)
    uh, uw = timed(lr.update, 50)

    def both():
        actor.step()
        lr.update()
    _, bw = timed(both, 50)
    print("| (%d, %d) | %d | %.2f | %.2f | %.2f | %.2f | %.2f |" % (n_agents, map_len, E, ah, aw, uh, uw, bw), flush=True)
