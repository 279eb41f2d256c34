#!/usr/bin/env python3
"""Tuning harness (not part of the product or the bench contract): times mapf_step variants selected
through the MAPF_STEP_* environment knobs, interleaved rounds in ONE process (guide rule 24)."""
import itertools
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mapf_rl_amd as M  # noqa: E402


def make(E, L, N, threads, pad, impl=0, seed=1, ablate=0):
    os.environ["MAPF_STEP_ABLATE"] = str(ablate)
    os.environ["MAPF_STEP_THREADS"] = str(threads)
    os.environ["MAPF_STEP_LDS_PAD"] = str(pad)
    os.environ["MAPF_STEP_IMPL"] = str(impl)
    maps, agents, goals, _ = M.generate_scenarios(E, L, N, 0.3, seed=seed)
    env = M.VecEnvironment(E, L, N)
    env.load(maps, agents, goals)
    return env


def time_env(env, tape, iters):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    for k in range(iters):
        env.step(tape[k % tape.shape[0]])
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / iters  # us per launch


def main():
    L, N = int(os.environ.get("TL", 32)), int(os.environ.get("TN", 40))
    variants = []
    for E in (4096, 16384):
        for threads, pad, impl in [(128, 0, 1), (64, 0, 0), (128, 0, 0), (64, 0, 10), (64, 0, 20), (64, 0, 30)]:
            if threads >= N:
                variants.append((E, threads, pad, impl))
    envs = {}
    tapes = {}
    for v in variants:
        E, threads, pad, impl = v
        envs[v] = make(E, L, N, threads, pad, impl % 10, ablate=impl // 10)
        if E not in tapes:
            tapes[E] = torch.randint(0, 5, (16, E, N), dtype=torch.int8, device="cuda")
    res = {v: [] for v in variants}
    for rnd in range(5):
        for v in variants:
            res[v].append(time_env(envs[v], tapes[v[0]], 50))
    alg = L * L + 821 * N + 1
    for v in variants:
        E, threads, pad, impl = v
        med = float(np.median(res[v]))
        print("impl=%d E=%5d threads=%3d pad=%5d  med %.2f us  min %.2f us  -> %.0f GB/s alg, %.2f ns/env" % (
            impl, E, threads, pad, med, min(res[v]), alg * E / med / 1e3, med * 1e3 / E))


if __name__ == "__main__":
    main()
