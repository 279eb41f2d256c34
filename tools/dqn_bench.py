#!/usr/bin/env python3
"""Measures the model-side rates at the BASELINE config-2 shape (not part of the bench contract):
actor-loop iteration (4096 envs x 40 agents through Network.step_batch + env.step) and learner updates
(192 x 18 x 40 windows)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mapf_rl_amd as M  # noqa: E402
from mapf_rl_amd.actor import VecActor  # noqa: E402
from mapf_rl_amd.learner import Learner  # noqa: E402
from mapf_rl_amd.replay import GlobalBuffer  # noqa: E402

E, L, N = int(os.environ.get("TE", 4096)), 32, 40
B = int(os.environ.get("TB", 192))
torch.backends.cudnn.benchmark = True
maps, agents, goals, _ = M.generate_scenarios(E, L, N, 0.3, seed=1)
env = M.VecEnvironment(E, L, N)
env.load(maps, agents, goals)
buf = GlobalBuffer(int(os.environ.get("TCAP", 512)), max_agents=N)
lr = Learner(buf, device="cuda", batch_size=B)
actor = VecActor(env, lr.model, buf, max_steps=int(os.environ.get("TMAX", 24)), seed=0, density=0.3)
t0 = time.time()
actor.step()
torch.cuda.synchronize()
print("first actor iteration (incl. MIOpen find) %.2fs" % (time.time() - t0), flush=True)
for k in range(3):
    t0 = time.time()
    actor.step()
    torch.cuda.synchronize()
    dt = time.time() - t0
    print("actor iteration %.1f ms -> %.0f env-steps/s (full loop), mem %.1f GB" % (dt * 1e3, E / dt, torch.cuda.max_memory_allocated() / 2**30), flush=True)
while len(buf) < 4000:
    actor.step()
torch.cuda.synchronize()
print("buffer size", len(buf), "episodes", actor.episodes, flush=True)
t0 = time.time()
lr.update()
torch.cuda.synchronize()
print("first update %.2fs" % (time.time() - t0), flush=True)
for k in range(3):
    t0 = time.time()
    out = lr.update()
    torch.cuda.synchronize()
    dt = time.time() - t0
    print("update %.1f ms -> %.2f updates/s  loss %.4f  mem %.1f GB" % (dt * 1e3, 1 / dt, float(out["loss"]), torch.cuda.max_memory_allocated() / 2**30), flush=True)
