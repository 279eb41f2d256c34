#!/usr/bin/env python3
"""_tall_tn (mapf_rl_amd/model.py): a^T b for K rows in the 10^5..10^6 range as a batched GEMM over row slabs + fp32 sum.
Which slab height at the 128-agent learner shapes (K = 393,216 and 786,432 rows)?"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
from mapf_rl_amd.model import _tall_tn
torch.manual_seed(0)
for K in (122880, 245760, 393216, 786432):
    for (m, n) in ((768, 256), (384, 256), (64, 128), (768, 64), (768, 784)):
        a = torch.randn((K, m), device="cuda", dtype=torch.bfloat16) * 0.1
        b = torch.randn((K, n), device="cuda", dtype=torch.bfloat16) * 0.1
        ref = None
        line = "K=%7d m=%3d n=%3d:" % (K, m, n)
        for rows in (4096, 8192, 16384, 32768, 65536):
            if K % rows:
                continue
            out = _tall_tn(a, b, rows=rows)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                out = _tall_tn(a, b, rows=rows)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 5
            if ref is None:
                ref = out
            err = float((out - ref).abs().max() / ref.abs().max())
            line += "  rows %5d: %.3f ms (%.0f TF, d %.1e)" % (rows, dt * 1e3, 2.0 * K * m * n / dt / 1e12, err)
        print(line, flush=True)
import torch.nn.functional as F
for R in (122880, 393216):
    dy = torch.randn((R, 768), device="cuda", dtype=torch.bfloat16) * 0.1
    x = torch.randn((R, 784), device="cuda", dtype=torch.bfloat16) * 0.1
    w = torch.randn((768, 784), device="cuda", dtype=torch.bfloat16) * 0.1
    wt = w.t().contiguous()
    for name, fn in (("fwd  x W^T          ", lambda: F.linear(x, w)), ("dx   dy (W^T)^T [TN]", lambda: F.linear(dy, wt)), ("dx   dy W      [NN]", lambda: torch.mm(dy, w))):
        fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): fn()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        print("R=%d %s %.3f ms (%.0f TF)" % (R, name, dt * 1e3, 2.0 * R * 768 * 784 / dt / 1e12), flush=True)
