#!/usr/bin/env python3
"""The merged multi-handle env step (mapf_multi_step: all curriculum levels in ONE launch) against one launch per level: time per step of
the whole level set and the fraction of the 8 TB/s HBM roofline on the AGGREGATE algorithmic bytes (L^2 + 821 N + 1 per
environment step), for growing numbers of environments per level.  Usage: multi_sweep.py [envs per level ...]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mapf_rl_amd as M  # noqa: E402
from bench import heuristic_actions  # noqa: E402
from mapf_rl_amd.environment import MultiEnvironment  # noqa: E402

LEVELS = [(4, 15), (3, 20), (2, 25), (6, 15), (5, 20), (1, 30), (4, 25)]  # the level set of tools/curriculum_iter.py
T = 40


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [512, 2048, 8192, 32768]
    print("| envs per level | levels | merged launch us | per-level launches us (sum of %d) | aggregate algorithmic MB | merged TB/s | frac of 8 TB/s | per-level frac |" % len(LEVELS))
    print("|---|---|---|---|---|---|---|---|")
    for E in sizes:
        envs, acts, bits, masks, tapes = [], [], [], [], []
        gen = torch.Generator(device="cuda").manual_seed(1)
        for N, L in LEVELS:
            env = M.VecEnvironment(E, L, N)
            env.reset_envs(None, 0.3, seed=N * 100 + L)
            env.check_status()
            envs.append(env)
            acts.append(torch.zeros((E, N), dtype=torch.int8, device="cuda"))
            bits.append(torch.zeros((E, env.obs_bits_row_dwords), dtype=torch.int32, device="cuda"))
            masks.append(torch.zeros(E, dtype=torch.uint8, device="cuda"))
            obs, _ = env.observe()
            tape = torch.empty((T, E, N), dtype=torch.int8, device="cuda")
            for t in range(T):
                tape[t] = heuristic_actions(obs, gen)
                obs, *_ = env.step(tape[t])
            tapes.append(tape)
        multi = MultiEnvironment(envs, acts, bits, masks)
        res = {}
        for mode in ("merged", "per level"):
            ts = []
            for rnd in range(3):
                evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(T)]
                torch.cuda.synchronize()
                for t in range(T):
                    for a, tape in zip(acts, tapes):
                        a.copy_(tape[t])       # (the actions land in the buffers the set reads; outside the timed bracket)
                    evs[t][0].record()         # HIP events around the step launch(es) of this iteration only
                    if mode == "merged":
                        multi.step()
                    else:
                        for env, a, b in zip(envs, acts, bits):
                            env.step(a, obs_bits_out=b)
                    evs[t][1].record()
                torch.cuda.synchronize()
                ts.append(sum(x.elapsed_time(y) for x, y in evs) * 1e3 / T)
            res[mode] = sorted(ts)[1]
        m, p = res["merged"], res["per level"]
        alg = sum(L * L + 821 * N + 1 for N, L in LEVELS) * E
        print("| %d | %d | %.1f | %.1f | %.1f | %.2f | %.3f | %.3f |" % (E, len(LEVELS), m, p, alg / 1e6, alg / m / 1e6, alg / m / 8e6, alg / p / 8e6), flush=True)
        for env in envs:
            env.check_status()
        del multi, envs, acts, bits, masks, tapes
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
