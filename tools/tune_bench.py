#!/usr/bin/env python3
"""Tuning harness on the BENCH workload: records the 80/20 heuristic action tape once (like bench.py), then
times env_step_kernel variants (MAPF_STEP_* knobs) replaying it, interleaved rounds in one process."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mapf_rl_amd as M  # noqa: E402
from bench import heuristic_actions  # noqa: E402

E, L, N, T = 4096, 32, 40, 120
maps, agents, goals, _ = M.generate_scenarios(E, L, N, 0.3, seed=1000)
agents_dev = torch.from_numpy(agents).cuda()


def make(threads):
    os.environ["MAPF_STEP_THREADS"] = str(threads)
    env = M.VecEnvironment(E, L, N)
    env.load(maps, agents, goals)
    return env


variants = [int(v) for v in os.environ.get("TV", "64,128").split(",")]
envs = {v: make(v) for v in variants}
gen = torch.Generator(device="cuda")
gen.manual_seed(77)
e0 = envs[variants[0]]
tape = torch.empty((T, E, N), dtype=torch.int8, device="cuda")
obs, pos = e0.observe()
for t in range(T):
    tape[t] = heuristic_actions(obs, gen)
    obs, pos, *_ = e0.step(tape[t])
res = {v: [] for v in variants}
for rnd in range(5):
    for v in variants:
        env = envs[v]
        env.set_agents(agents_dev)
        for t in range(20):
            env.step(tape[t])
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        s.record()
        for t in range(20, T):
            env.step(tape[t])
        e.record()
        torch.cuda.synchronize()
        res[v].append(s.elapsed_time(e) * 1e3 / (T - 20))
alg = (L * L + 821 * N + 1) * E
for v in variants:
    med = float(np.median(res[v]))
    print("threads=%3d  med %.2f us  min %.2f us  -> %.0f GB/s alg (frac %.3f)" % (v, med, min(res[v]), alg / med / 1e3, alg / med / 1e3 / 8000))
