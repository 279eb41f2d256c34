#!/usr/bin/env python3
"""Times mapf_recurrent_forward_save and mapf_recurrent_backward (dense rows, random data) at a learner shape: recur_bwd_time.py [T E N].
Environment switches of the kernels (e.g. MAPF_RBWD_STAGGER_SLOTS / _UNITS) are read once per process: one run per setting."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mapf_rl_amd import fused  # noqa: E402
from mapf_rl_amd._lib import check, lib  # noqa: E402

T, E, N = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (18, 192, 40)
dev, bf = torch.device("cuda"), torch.bfloat16
g = torch.Generator(device="cuda").manual_seed(0)
w = (torch.randn(fused.RECUR_WEIGHT_ELEMS, device=dev, generator=g) * 0.05).to(bf)
wt = (torch.randn(fused.RECUR_WEIGHT_ELEMS, device=dev, generator=g) * 0.05).to(bf)
bias = torch.randn(fused.RECUR_BIAS_ELEMS, device=dev, generator=g) * 0.1
gi = (torch.randn((T, E, N, 768), device=dev, generator=g) * 0.5).to(bf)
h0 = (torch.randn((E, N, 256), device=dev, generator=g) * 0.3).to(bf)
comm = ((torch.rand((T, E, N, N), device=dev, generator=g) < 0.1) | torch.eye(N, dtype=torch.bool, device=dev)).to(torch.uint8).contiguous()
d_a0 = (torch.randn((T, E, 256), device=dev, generator=g) * 0.1).to(bf)
R = T * E * N
saves = [torch.zeros(s_, dtype=bf, device=dev) for s_ in ((R, 256), (R, 1024), (2, R, 256), (2, R, 384), (2, R, 128), (2, R, 64), (2, R, 1024), (2, T * E, 2, 48, 64))]
outs = [torch.zeros(s_, dtype=bf, device=dev) for s_ in ((R, 768), (R, 768), (2, R, 768), (2, R, 768), (2, R, 64), (2, R, 384))]
outs.append(torch.zeros((E, 2432), dtype=torch.float32, device=dev))
h_out, a0 = torch.zeros((E, N, 256), dtype=bf, device=dev), torch.zeros((T, E, 256), dtype=bf, device=dev)
sp = (ctypes.c_void_p * 8)(*[t.data_ptr() for t in saves])
op = (ctypes.c_void_p * 7)(*[t.data_ptr() for t in outs])
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: ctypes.c_void_p(t.data_ptr())
fwd = lambda: check(lib.mapf_recurrent_forward_save(p(gi), p(h0), p(comm), p(w), p(bias), T, E, N, p(h_out), p(a0), sp, None, 0, st), "save")
bwd = lambda: check(lib.mapf_recurrent_backward(sp, p(comm), p(d_a0), p(wt), T, E, N, op, None, 0, st), "bwd")


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return best


tf, tb = timed(fwd), timed(bwd)
chk = sum(float(o.float().abs().sum()) for o in outs)
print("T=%d E=%d N=%d  %s  forward_save %.3f ms  backward %.3f ms  (checksum %.6e)" % (
    T, E, N, " ".join("%s=%s" % (k, v) for k, v in sorted(os.environ.items()) if k.startswith("MAPF_R")), tf, tb, chk), flush=True)
