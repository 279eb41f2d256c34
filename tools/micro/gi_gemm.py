#!/usr/bin/env python3
"""The input projection GEMM of the actor path ([rows, 784] x W_ih^T [784, 768], bf16): library kernel time against operand layout and
row chunking (fused.mm_rows chunks the rows: a stream-K kernel hung on a shared GPU in round 3).  Run on the GPU."""
import torch

dev = "cuda"
w = (torch.randn(768, 784, device=dev) * 0.05).to(torch.bfloat16)
wt = w.t().contiguous()


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
    return e0.elapsed_time(e1) / n


for rows in (12544, 32768, 163840):
    x = (torch.randn(rows, 784, device=dev) * 0.5).to(torch.bfloat16)
    out = torch.empty(rows, 768, dtype=torch.bfloat16, device=dev)
    fl = rows * 784 * 768 * 2
    res = []
    for name, fn in (("x @ w.t()", lambda: torch.mm(x, w.t(), out=out)), ("x @ wt (contiguous)", lambda: torch.mm(x, wt, out=out)),
                     ("linear", lambda: torch.nn.functional.linear(x, w))):
        ms = timed(fn)
        res.append("%s %.3f ms (%.2f PF/s)" % (name, ms, fl / ms / 1e12))
    for chunk in (16384, 32768, 65536):
        if chunk < rows:
            def f():
                for i in range(0, rows, chunk):
                    torch.mm(x[i:i + chunk], w.t(), out=out[i:i + chunk])
            ms = timed(f)
            res.append("chunks of %d: %.3f ms (%.2f PF/s)" % (chunk, ms, fl / ms / 1e12))
    print("rows %6d: " % rows + " | ".join(res), flush=True)
from mapf_rl_amd.fused import mm_rows  # noqa: E402

x = (torch.randn(163840, 784, device=dev) * 0.5).to(torch.bfloat16)
print("mm_rows(163840 rows): %.3f ms" % timed(lambda: mm_rows(x, w)))
