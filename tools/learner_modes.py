#!/usr/bin/env python3
"""Learner.update wall time with and without the side-stream prefetch (config 2 shape).  Round 2 also tried sampling one update
early + a high-priority update stream: 39.7 ms against 39.5 (off) and 38.7 (on): the update's own kernels saturate the chip."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from mapf_rl_amd.learner import Learner
from mapf_rl_amd.replay import GlobalBuffer
N = int(os.environ.get("NAGENTS", 40))
dev = torch.device("cuda")
def make():
    buf = GlobalBuffer(64, max_agents=N, device=dev)
    g2 = torch.Generator(device=dev); g2.manual_seed(5)
    RD, CW, S = buf.row_dwords, (N + 31) // 32, 96
    for k in range(64):
        td = torch.zeros(256, dtype=torch.float64, device=dev); td[:S] = torch.rand(S, generator=g2, device=dev, dtype=torch.float64) + 0.05
        buf.add_episode_device(N, S, k % 2, torch.randint(-2**31, 2**31 - 1, (S + 1, RD), generator=g2, device=dev, dtype=torch.int32) &
                               torch.randint(-2**31, 2**31 - 1, (S + 1, RD), generator=g2, device=dev, dtype=torch.int32),
                               torch.randint(0, 2**20, (S + 1, N, CW), generator=g2, device=dev, dtype=torch.int32),
                               torch.randint(0, 5, (S,), generator=g2, device=dev, dtype=torch.uint8),
                               (torch.rand(S, generator=g2, device=dev) - 0.5).half(), (torch.randn((S, 256), generator=g2, device=dev) * 0.3).half(), td)
    return buf
res = {}
for rnd in range(2):
    for mode in (False, True):
        torch.manual_seed(0)
        lr = Learner(make(), device=dev, batch_size=192, prefetch=mode)
        for _ in range(3):
            lr.update()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        K = 10
        for _ in range(K):
            out = lr.update()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / K
        res.setdefault(str(mode), []).append(dt * 1e3)
        assert torch.isfinite(out["loss"])
        del lr
for k, v in res.items():
    print("prefetch=%-6s  %s ms per update" % (k, ["%.2f" % x for x in v]), flush=True)
