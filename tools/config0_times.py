#!/usr/bin/env python3
"""BASELINE configs[0] on the GPU ("16x16 grid, 40 agents, rho = 0.3 (test16_40_0.3.pkl), 1 env"; SURVEY.md 8(d) reads it two ways:
C1a = the fixture as it is, 40x40 / 16 agents, 200 cases; C1b = BASELINE's literal 16x16 / 40 agents, generated):

  * ONE environment: `mapf_step` (fused step + observe) launched back to back (HIP events: what the kernel costs at a launch of one
    workgroup) and launch + synchronise per step (what a caller that needs every observation on the host before the next action
    sees), then the same through the reference-compatible single-environment facade (`mapf_rl_amd.Environment.step`: numpy in / out);
  * the reference's own use of the fixture (test.py:105-143): the 200 cases stepped in lock-step with the network in the loop
    (`evaluate.evaluate`, random-init weights, 256 steps, no early exit unless every case finished), cases/s and env-steps/s.

The CPU columns of the same table (the unmodified reference and the C oracle on one host core) come from oracle/time_reference.py in the
build container (profiles/r06_config0_reference_cpu.txt).  Usage: python tools/config0_times.py > profiles/r06_config0_gpu.txt"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mapf_rl_amd as M  # noqa: E402
from bench import heuristic_actions  # noqa: E402
from mapf_rl_amd.evaluate import evaluate, load_fixture_npz  # noqa: E402
from mapf_rl_amd.model import Network  # noqa: E402

dev = torch.device("cuda")
FIX = os.path.join(ROOT, "tests", "golden", "fixture_scenarios.npz")


def scenarios(tag):
    if tag == "C1a":
        t = load_fixture_npz(FIX, 16)
        return (np.stack(t["maps"]).astype(np.int8), np.stack(t["agents"]).astype(np.int16), np.stack(t["goals"]).astype(np.int16),
                "C1a: test16_40_0.3.pkl, 40x40 grid, 16 agents")
    maps, agents, goals, redraws = M.generate_scenarios(64, 16, 40, 0.3, seed=2024)
    return maps, agents, goals, "C1b: 16x16 grid, 40 agents, rho 0.3 (generated; %d infeasible draws skipped)" % redraws


def single_env(tag, T=200, reps=20):
    maps, agents, goals, name = scenarios(tag)
    L, N = maps.shape[1], agents.shape[1]
    env = M.VecEnvironment(1, L, N, device=dev)
    env.load(maps[:1], agents[:1], goals[:1])
    gen = torch.Generator(device=dev).manual_seed(5)
    tape = torch.empty((T, 1, N), dtype=torch.int8, device=dev)
    obs, pos = env.observe()
    for t in range(T):
        tape[t] = heuristic_actions(obs, gen)
        obs, pos, *_ = env.step(tape[t])
    a0 = torch.from_numpy(agents[:1]).to(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    b2b = []
    for r in range(reps + 2):
        env.set_agents(a0)
        e0.record()
        for t in range(T):
            env.step(tape[t])
        e1.record()
        torch.cuda.synchronize()
        if r >= 2:
            b2b.append(e0.elapsed_time(e1) * 1e3 / T)
    sync = []
    for r in range(5):
        env.set_agents(a0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for t in range(T):
            env.step(tape[t])
            torch.cuda.synchronize()
        sync.append((time.perf_counter() - t0) * 1e6 / T)
    # the reference-compatible facade: numpy observations back on the host every step
    fe = M.Environment(map_length=L, num_agents=N)
    fe.load(maps[0].astype(np.float32), agents[0].astype(int), goals[0].astype(int))
    acts = tape[:, 0].cpu().numpy().astype(int).tolist()
    fac = []
    for r in range(4):
        fe.load(maps[0].astype(np.float32), agents[0].astype(int), goals[0].astype(int))
        t0 = time.perf_counter()
        for t in range(T):
            fe.step(acts[t])
        if r:
            fac.append((time.perf_counter() - t0) * 1e6 / T)
    alg = L * L + 821 * N + 1
    print("%s, ONE environment (%d algorithmic bytes per step):" % (name, alg))
    print("    mapf_step back to back (HIP events):        median %.2f us per step = %.0f env-steps/s   (launch-bound: %.4f of the 8 TB/s roofline)" % (
        float(np.median(b2b)), 1e6 / float(np.median(b2b)), alg / (float(np.median(b2b)) * 1e-6) / 8e12))
    print("    mapf_step + synchronise per step:            median %.1f us per step = %.0f env-steps/s" % (float(np.median(sync)), 1e6 / float(np.median(sync))))
    print("    Environment.step facade (numpy in / out):    median %.1f us per step = %.0f env-steps/s" % (float(np.median(fac)), 1e6 / float(np.median(fac))), flush=True)


def lockstep_eval():
    torch.manual_seed(0)
    net = Network().to(dev).eval()
    for n in (16, 32, 64):
        tests = load_fixture_npz(FIX, n)
        evaluate(net, tests, dev, max_steps=8)  # warm-up (packs, allocator)
        torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            f, ms, steps, ok = evaluate(net, tests, dev)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        dt = float(np.median(ts))
        print("test%d_40_0.3 (200 cases x 40x40 / %d agents), lock-step evaluate(), random-init network, 256 steps: %.2f s = %.0f cases/s = %.3g env-steps/s "
              "(finish rate %.3f)" % (n, n, dt, 200 / dt, 200 * 256 / dt, f), flush=True)


if __name__ == "__main__":
    print("GPU: %s" % torch.cuda.get_device_name(0))
    single_env("C1a")
    single_env("C1b")
    lockstep_eval()
