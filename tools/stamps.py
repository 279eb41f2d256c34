#!/usr/bin/env python3
"""Diagnostic: per-block phase stamps (s_memtime) of env_step_kernel; prints mean phase durations in cycles."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mapf_rl_amd as M  # noqa: E402
from mapf_rl_amd._lib import lib  # noqa: E402

E, L, N = int(os.environ.get("TE", 4096)), int(os.environ.get("TL", 32)), int(os.environ.get("TN", 40))
maps, agents, goals, _ = M.generate_scenarios(E, L, N, 0.3, seed=1)
env = M.VecEnvironment(E, L, N)
env.load(maps, agents, goals)
tape = torch.randint(0, 5, (8, E, N), dtype=torch.int8, device="cuda")
for k in range(4):
    env.step(tape[k])
buf = torch.zeros((E, 8), dtype=torch.int64, device="cuda")
lib.mapf_debug_set_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
lib.mapf_debug_set_stamps(env._h, ctypes.c_void_p(buf.data_ptr()))
env.step(tape[5])
torch.cuda.synchronize()
b = buf.cpu().numpy().astype(np.float64)
t0 = b[:, 0].min()
names = ["start", "loads+init->barrierA", "step logic", "fields (navi wait+deposit)", "expand+store issue", "store drain"]
print("blocks", E, "kernel span (cycles, s_memtime 100MHz?)", b[:, 5].max() - t0)
for k in range(6):
    col = b[:, k] - t0
    print("stamp %d %-28s mean %9.1f  min %9.1f  max %9.1f   phase mean %8.1f" % (
        k, names[k], col.mean(), col.min(), col.max(), (b[:, k] - b[:, k - 1]).mean() if k else 0))

# how the blocks' start / end times spread over the launch (dispatch rate vs per-block latency)
st, en = np.sort(b[:, 0] - t0), np.sort(b[:, 5] - t0)
q = [0, 10, 25, 50, 75, 90, 100]
print("block START percentiles (cycles):", " ".join("%d%%:%d" % (k, np.percentile(st, k)) for k in q))
print("block END   percentiles (cycles):", " ".join("%d%%:%d" % (k, np.percentile(en, k)) for k in q))
print("per-block duration (cycles): mean %.0f  min %.0f  max %.0f" % ((b[:, 5] - b[:, 0]).mean(), (b[:, 5] - b[:, 0]).min(), (b[:, 5] - b[:, 0]).max()))
