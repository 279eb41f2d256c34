"""Timing of the GRU input projection's backward GEMMs at the learner's shape (R = 192*16*40 rows):
d_x = d_y W as the NN GEMM autograd issues vs the TN form on a transposed copy of W; dW = d_y^T x as one GEMM vs split-K."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from mapf_rl_amd.model import _tall_tn

R = 192 * 16 * 40
dev = "cuda"
dy = torch.randn(R, 768, device=dev, dtype=torch.bfloat16)
x = torch.randn(R, 784, device=dev, dtype=torch.bfloat16)
w = torch.randn(768, 784, device=dev, dtype=torch.bfloat16)
wt = w.t().contiguous()


def t(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


print("fwd  F.linear(x, w)            %.0f us" % t(lambda: F.linear(x, w)))
print("dx   dy @ w (NN)               %.0f us" % t(lambda: torch.mm(dy, w)))
print("dx   F.linear(dy, w^T copy)    %.0f us" % t(lambda: F.linear(dy, wt)))
print("dx   transpose copy + linear   %.0f us" % t(lambda: F.linear(dy, w.t().contiguous())))
xp = torch.zeros(R, 896, device=dev, dtype=torch.bfloat16)
wp = torch.zeros(768, 896, device=dev, dtype=torch.bfloat16)
print("dx   dy @ w padded to 896      %.0f us" % t(lambda: torch.mm(dy, wp)))
print("dW   dy^T @ x (one GEMM)       %.0f us" % t(lambda: torch.mm(dy.t(), x)))
for rows in (4096, 8192, 16384, 30720):
    print("dW   _tall_tn rows=%-6d       %.0f us" % (rows, t(lambda: _tall_tn(dy, x, rows))))
