"""How to add 30,720 per-workgroup bias partials [7, 30720, 128] fp32: one reduction vs two stages vs a GEMM with ones."""
import time, torch
x = torch.randn(7, 30720, 128, device="cuda")
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
print("sum(dim=1)                 %.0f us" % t(lambda: x.sum(dim=1)))
for c in (64, 128, 256, 512):
    print("two stages, inner %-4d      %.0f us" % (c, t(lambda: x.view(7, -1, c, 128).sum(dim=2).sum(dim=1))))
ones = torch.ones(1, 30720, device="cuda")
print("ones @ x                   %.0f us" % t(lambda: torch.matmul(ones, x)))
print("max diff two-stage vs one  %.2e" % float((x.view(7, -1, 256, 128).sum(2).sum(1) - x.sum(1)).abs().max()))
