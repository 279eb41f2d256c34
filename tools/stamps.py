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
names = ["start", "loads+init->barrierA", "step logic", "fields (navi wait+deposit)", "expand+store issue", "store drain"]
G = max(1, E // max(1, int((b[:, 5] != 0).sum())))  # environments per workgroup: only the first E / G rows of the buffer are written
b = b[: E // G]
print("environments", E, "workgroups", b.shape[0])
for k in range(1, 6):
    print("phase %d %-28s mean %8.1f cycles" % (k, names[k], (b[:, k] - b[:, k - 1]).mean()))

# (absolute times are not comparable between blocks: every XCD has its own counter; only differences inside a block are used)
print("per-block duration (cycles): mean %.0f  min %.0f  max %.0f" % ((b[:, 5] - b[:, 0]).mean(), (b[:, 5] - b[:, 0]).min(), (b[:, 5] - b[:, 0]).max()))
print("G = %d; start -> both load rounds issued: %.0f; issued -> LDS init done + barrier: %.0f; step logic: %.0f; navi still outstanding after the step logic: %.0f; "
      "field deposit: %.0f; expand + store issue: %.0f; store drain: %.0f" % (G, (b[:, 6] - b[:, 0]).mean(), (b[:, 1] - b[:, 6]).mean(), (b[:, 2] - b[:, 1]).mean(),
      (b[:, 7] - b[:, 2]).mean(), (b[:, 3] - b[:, 7]).mean(), (b[:, 4] - b[:, 3]).mean(), (b[:, 5] - b[:, 4]).mean()))
