#!/usr/bin/env python3
"""The recurrence's weight-gradient GEMMs (out[m, n] = a^T b, a [K, m], b [K, n], K = 26-53 k rows, fp32 out): library time against the
way K is split (update._tall_tn_into: bmm over chunks of `rows` + a sum).  Run on the GPU."""
import torch

dev = "cuda"


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def split(a, b, out, rows):
    K, m = a.shape
    S = K // rows
    part = torch.bmm(a[:S * rows].view(S, rows, m).transpose(1, 2), b[:S * rows].view(S, rows, -1), out_dtype=torch.float32)
    torch.sum(part, dim=0, out=out)


for K, m, n in ((53248, 768, 256), (26624, 768, 256), (53248, 384, 256), (53248, 768, 64), (53248, 64, 128), (12288, 768, 256)):
    a = (torch.randn(K, m, device=dev) * 0.1).to(torch.bfloat16)
    b = (torch.randn(K, n, device=dev) * 0.1).to(torch.bfloat16)
    out = torch.empty(m, n, dtype=torch.float32, device=dev)
    ref = a.float().t() @ b.float()
    res = []
    for rows in (1024, 2048, 4096, 8192):
        us = timed(lambda: split(a, b, out, rows))
        res.append("split %d: %.0f us" % (rows, us))
    us = timed(lambda: out.copy_(torch.mm(a.t(), b, out_dtype=torch.float32)))
    err = ((torch.mm(a.t(), b, out_dtype=torch.float32) - ref).abs().max() / ref.abs().max()).item()
    res.append("one mm: %.0f us (rel err %.1e)" % (us, err))
    us = timed(lambda: out.copy_(torch.mm(b.t(), a, out_dtype=torch.float32).t()))
    res.append("one mm, transposed product: %.0f us" % us)
    fl = 2.0 * K * m * n
    print("K=%6d m=%4d n=%4d (%.1f GFLOP, %.0f MB): " % (K, m, n, fl / 1e9, (K * (m + n) * 2) / 1e6) + " | ".join(res), flush=True)
