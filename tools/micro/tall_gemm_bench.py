#!/usr/bin/env python3
"""mapf_tall_tn / mapf_latent_grad_rows / mapf_input_proj_rows against the library formulations rounds 1-4 used (bmm + sum in fp32,
torch.mm), at the learner's shapes (config 2: ~26 k compact rows, ~12 k distinct observations; 6 agents: 2 k / 6 k)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from mapf_rl_amd.fused import tall_tn_into  # noqa: E402


def old_tall(out, a, b, rows=8192):
    K, m = a.shape
    S = K // rows
    if S > 1:
        part = torch.bmm(a[:S * rows].view(S, rows, m).transpose(1, 2), b[:S * rows].view(S, rows, -1), out_dtype=torch.float32)
        torch.sum(part, dim=0, out=out)
        if K > S * rows:
            out += torch.mm(a[S * rows:].t(), b[S * rows:], out_dtype=torch.float32)
    else:
        out.copy_(torch.mm(a.t(), b, out_dtype=torch.float32))


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


g = torch.Generator(device="cuda").manual_seed(0)
print("%-44s %10s %10s %8s" % ("K x m x n", "own us", "library us", "TFLOP/s"))
for K, m, n, dt, rows in ((26624, 768, 256, torch.bfloat16, 2048), (53248, 384, 256, torch.bfloat16, 2048), (53248, 64, 128, torch.bfloat16, 2048),
                          (53248, 768, 64, torch.bfloat16, 2048), (53248, 768, 256, torch.bfloat16, 2048), (12288, 768, 784, torch.bfloat16, 4096),
                          (12288 * 49, 16, 128, torch.float16, 8192), (2048, 768, 256, torch.bfloat16, 2048), (4096, 768, 256, torch.bfloat16, 2048),
                          (6144, 768, 784, torch.bfloat16, 4096)):
    a = (torch.randn((K, m), device="cuda", generator=g) * 0.5).to(dt)
    b = (torch.randn((K, n), device="cuda", generator=g) * 0.5).to(dt)
    out = torch.empty((m, n), device="cuda")
    own = timed(lambda: tall_tn_into(out, a, b))
    lib = timed(lambda: old_tall(out, a, b, rows))
    print("%-44s %10.1f %10.1f %8.1f" % ("%d x %d x %d (%s)" % (K, m, n, str(dt)[6:]), own, lib, 2.0 * K * m * n / own / 1e6), flush=True)

# ---- the other pieces ----
import ctypes  # noqa: E402

from mapf_rl_amd._lib import check, lib  # noqa: E402
from mapf_rl_amd.fused import INPROJ_PACKED_ELEMS, LATGRAD_PACKED_ELEMS, input_proj_rows, latent_grad_rows, sum_parts_into  # noqa: E402

st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
w = torch.randn((768, 784), device="cuda", generator=g) * 0.05
wb = w.to(torch.bfloat16)
pk, pkt = torch.empty(INPROJ_PACKED_ELEMS, dtype=torch.bfloat16, device="cuda"), torch.empty(LATGRAD_PACKED_ELEMS, dtype=torch.bfloat16, device="cuda")
check(lib.mapf_input_proj_pack(ctypes.c_void_p(w.data_ptr()), ctypes.c_void_p(pk.data_ptr()), st))
check(lib.mapf_latent_grad_pack(ctypes.c_void_p(w.data_ptr()), ctypes.c_void_p(pkt.data_ptr()), st))
print("%-44s %10s %10s" % ("rows", "own us", "library us"))
for rows in (2048, 6144, 12288, 24576):
    lat = (torch.randn((rows, 784), device="cuda", generator=g) * 0.5).to(torch.bfloat16)
    dgi = (torch.randn((rows, 768), device="cuda", generator=g) * 0.5).to(torch.bfloat16)
    o1, o2 = torch.empty((rows, 768), dtype=torch.bfloat16, device="cuda"), torch.empty((rows, 784), dtype=torch.bfloat16, device="cuda")
    print("%-44s %10.1f %10.1f" % ("input projection, %d rows" % rows, timed(lambda: input_proj_rows(lat, pk, out=o1)), timed(lambda: torch.mm(lat, wb.t(), out=o1))))
    print("%-44s %10.1f %10.1f" % ("latent gradient, %d rows" % rows, timed(lambda: latent_grad_rows(dgi, pkt, out=o2)), timed(lambda: torch.mm(dgi, wb, out=o2))), flush=True)
parts = torch.randn((128, 128, 3, 3, 128), device="cuda", generator=g)
out = torch.empty((128, 3, 3, 128), device="cuda")
print("%-44s %10.1f %10.1f" % ("sum of 128 weight-gradient slabs", timed(lambda: sum_parts_into([out], [parts])), timed(lambda: torch.sum(parts, dim=0, out=out))))
