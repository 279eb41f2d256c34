#!/usr/bin/env python3
"""env_step_kernel time at the other BASELINE shapes (parity-test configs; not bench lines)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import mapf_rl_amd as M
from bench import heuristic_actions
for (E, L, N) in [(4096, 32, 40), (4096, 64, 40), (2048, 64, 128), (4096, 40, 16), (4096, 16, 40), (8192, 20, 6)]:
    env = M.VecEnvironment(E, L, N)
    env.reset_envs(None, 0.3, seed=1)
    env.check_status()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); env.reset_envs(None, 0.3, seed=2); e.record(); torch.cuda.synchronize()
    t_reset = s.elapsed_time(e) * 1e3
    gen = torch.Generator(device="cuda"); gen.manual_seed(1)
    T = 60
    tape = torch.empty((T, E, N), dtype=torch.int8, device="cuda")
    obs, pos = env.observe()
    for t in range(T):
        tape[t] = heuristic_actions(obs, gen); obs, pos, *_ = env.step(tape[t])
    start = env.agents_pos().clone()
    ts = []
    for rnd in range(3):
        s.record()
        for t in range(T): env.step(tape[t])
        e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e) * 1e3 / T)
    alg = (L * L + 821 * N + 1) * E
    us = float(np.median(ts))
    print("E=%5d L=%2d N=%3d  step %.2f us  %.0f GB/s alg (frac %.3f)  %.1f M env-steps/s   on-device reset of all envs %.0f us" % (
        E, L, N, us, alg / us / 1e3, alg / us / 8e6, E / us, t_reset), flush=True)
