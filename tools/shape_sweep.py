#!/usr/bin/env python3
"""env_step_kernel time at the BASELINE shapes and the small curriculum shapes (parity-test configs; not bench lines).

    python3 tools/shape_sweep.py                       # HIP-event timing, one line per shape
    rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 tools/shape_sweep.py
    python3 tools/shape_sweep.py --summarize DIR       # per-shape kernel averages from the trace (markdown)

Shapes are (E, L, N); E = 16384 at 32x32/40 puts the working set (navi 336 MB + observations 318 MB) beyond the 256 MiB
Infinity Cache, so FETCH/WRITE there are HBM traffic proper."""
import csv
import glob
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SHAPES = [(4096, 32, 40), (16384, 32, 40), (32768, 32, 40), (4096, 64, 40), (2048, 64, 128), (4096, 40, 16), (4096, 16, 40),
          (8192, 20, 6), (65536, 20, 6), (16384, 10, 1), (262144, 10, 1), (65536, 15, 3)]
T = 40


def alg_bytes(E, L, N):
    return (L * L + 821 * N + 1) * E


def run():
    import numpy as np
    import torch

    import mapf_rl_amd as M
    from bench import heuristic_actions

    shapes = SHAPES
    if len(sys.argv) > 1:
        shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
    for (E, L, N) in shapes:
        env = M.VecEnvironment(E, L, N)
        env.reset_envs(None, 0.3, seed=1)
        env.check_status()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        env.reset_envs(None, 0.3, seed=2)
        e.record()
        torch.cuda.synchronize()
        t_reset = s.elapsed_time(e) * 1e3
        gen = torch.Generator(device="cuda")
        gen.manual_seed(1)
        tape = torch.empty((T, E, N), dtype=torch.int8, device="cuda")
        obs, pos = env.observe()
        for t in range(T):
            tape[t] = heuristic_actions(obs, gen)
            obs, pos, *_ = env.step(tape[t])
        ts = []
        for rnd in range(3):
            s.record()
            for t in range(T):
                env.step(tape[t])
            e.record()
            torch.cuda.synchronize()
            ts.append(s.elapsed_time(e) * 1e3 / T)
        env.check_status()
        alg = alg_bytes(E, L, N)
        us = float(np.median(ts))
        print("SHAPE E=%d L=%d N=%d  step %.2f us  %.0f GB/s alg (frac %.3f)  %.1f M env-steps/s   on-device reset of all envs %.0f us" % (
            E, L, N, us, alg / us / 1e3, alg / us / 8e6, E / us, t_reset), flush=True)
        del env, tape, obs, pos
        torch.cuda.empty_cache()


def summarize(d, log=None):
    """Per-shape kernel averages from a kernel trace of run(): the shapes run one after the other and each launches the
    step+observe kernel T (recording pass) + 3 T (timed passes) times, so the step launches in start order split into runs of 4 T."""
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    assert files, "no *kernel_trace.csv under " + d
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            m = re.search(r"env_step_kernel<([^>]*)>", r["Kernel_Name"])
            if not m:
                continue
            targs = [a.strip() for a in m.group(1).split(",")]
            if targs[2] != "true" or targs[3] != "true":  # step + observe launches only
                continue
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), m.group(1),
                         int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"])))
    rows.sort()
    shapes = []
    if log and os.path.exists(log):
        for line in open(log):
            m = re.match(r"SHAPE E=(\d+) L=(\d+) N=(\d+)\s+step ([\d.]+) us", line)
            if m:
                shapes.append((int(m.group(1)), int(m.group(2)), int(m.group(3)), float(m.group(4))))
    assert len(rows) == 4 * T * len(shapes), (len(rows), len(shapes))
    print("| envs | grid | agents | instantiation `<W, R, step, obs, VEC, ITERS, NT, G>` | blocks | avg us (rocprofv3, %d timed launches) | min us | "
          "algorithmic MB per launch | TB/s | frac of 8 TB/s | HIP-event us per step (incl. launch gaps) |" % (3 * T))
    print("|---|---|---|---|---|---|---|---|---|---|---|")
    for k, (E, L, N, ev) in enumerate(shapes):
        grp = rows[4 * T * k:4 * T * (k + 1)]
        timed = [d for _, d, _, _ in grp[T:]]
        avg = sum(timed) / len(timed) / 1e3
        alg = alg_bytes(E, L, N)
        print("| %d | %dx%d | %d | `%s` | %d | %.2f | %.2f | %.1f | %.2f | %.3f | %.2f |" % (
            E, L, L, N, grp[-1][2], grp[-1][3], avg, min(timed) / 1e3, alg / 1e6, alg / avg / 1e6, alg / avg / 8e6, ev))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--summarize":
        summarize(sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None)
    else:
        run()
