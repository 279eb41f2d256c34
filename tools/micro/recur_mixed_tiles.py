#!/usr/bin/env python3
"""Probe (round-5 review, C2 learner): the learner launches its recurrence kernels with ONE compact width per batch, the batch's maximum
reachable-agent count -- at config 2 three agent tiles (48 rows) although 148 of 192 windows need <= 16 agents, 40 need 17..32 and 4 more
(gpurun_out/r03_v_update_times.txt).  What would per-window tile counts buy?  The existing kernels on a SPLIT batch -- 148 windows at 16
rows, 40 at 32, 4 at 48, three launches on three streams at the same time -- against the one launch of 192 windows at 48 rows, forward-save
and BPTT (dense rows, random data), plus every uniform width for reference."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mapf_rl_amd import fused  # noqa: E402
from mapf_rl_amd._lib import check, lib  # noqa: E402

T = 18
dev, bf = torch.device("cuda"), torch.bfloat16
g = torch.Generator(device="cuda").manual_seed(0)
w = (torch.randn(fused.RECUR_WEIGHT_ELEMS, device=dev, generator=g) * 0.05).to(bf)
wt = (torch.randn(fused.RECUR_WEIGHT_ELEMS, device=dev, generator=g) * 0.05).to(bf)
bias = torch.randn(fused.RECUR_BIAS_ELEMS, device=dev, generator=g) * 0.1
p = lambda t: ctypes.c_void_p(t.data_ptr())


class Part:
    def __init__(self, E, N, stream):
        self.E, self.N, self.stream = E, N, stream
        self.gi = (torch.randn((T, E, N, 768), device=dev, generator=g) * 0.5).to(bf)
        self.h0 = (torch.randn((E, N, 256), device=dev, generator=g) * 0.3).to(bf)
        self.comm = ((torch.rand((T, E, N, N), device=dev, generator=g) < 0.1) | torch.eye(N, dtype=torch.bool, device=dev)).to(torch.uint8).contiguous()
        self.d_a0 = (torch.randn((T, E, 256), device=dev, generator=g) * 0.1).to(bf)
        R = T * E * N
        self.saves = [torch.zeros(s_, dtype=bf, device=dev) for s_ in ((R, 256), (R, 1024), (2, R, 256), (2, R, 384), (2, R, 128), (2, R, 64), (2, R, 1024), (2, T * E, 2, 48, 64))]
        self.outs = [torch.zeros(s_, dtype=bf, device=dev) for s_ in ((R, 768), (R, 768), (2, R, 768), (2, R, 768), (2, R, 64), (2, R, 384))]
        self.outs.append(torch.zeros((E, 2432), dtype=torch.float32, device=dev))
        self.h_out, self.a0 = torch.zeros((E, N, 256), dtype=bf, device=dev), torch.zeros((T, E, 256), dtype=bf, device=dev)
        self.sp = (ctypes.c_void_p * 8)(*[t.data_ptr() for t in self.saves])
        self.op = (ctypes.c_void_p * 7)(*[t.data_ptr() for t in self.outs])

    def fwd(self):
        st = ctypes.c_void_p(self.stream.cuda_stream)
        check(lib.mapf_recurrent_forward_save(p(self.gi), p(self.h0), p(self.comm), p(w), p(bias), T, self.E, self.N, p(self.h_out), p(self.a0), self.sp, None, 0, st), "save")

    def bwd(self):
        st = ctypes.c_void_p(self.stream.cuda_stream)
        check(lib.mapf_recurrent_backward(self.sp, p(self.comm), p(self.d_a0), p(wt), T, self.E, self.N, self.op, None, 0, st), "bwd")


def timed(parts, which, reps=10):
    main = torch.cuda.current_stream()

    def once():
        for q in parts:
            if q.stream != main:
                q.stream.wait_stream(main)
            getattr(q, which)()
        for q in parts:
            if q.stream != main:
                main.wait_stream(q.stream)

    for _ in range(3):
        once()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            once()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return best


main = torch.cuda.current_stream()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
for E, N in ((192, 48), (192, 32), (192, 16)):
    q = [Part(E, N, main)]
    print("one launch, %3d windows x %2d rows:                     forward_save %.3f ms   backward %.3f ms" % (E, N, timed(q, "fwd"), timed(q, "bwd")), flush=True)
split = [Part(4, 48, main), Part(40, 32, s1), Part(148, 16, s2)]
print("three launches side by side, 4 x 48 + 40 x 32 + 148 x 16: forward_save %.3f ms   backward %.3f ms" % (timed(split, "fwd"), timed(split, "bwd")), flush=True)
split2 = [Part(44, 48, main), Part(148, 16, s2)]
print("two launches side by side, 44 x 48 + 148 x 16:            forward_save %.3f ms   backward %.3f ms" % (timed(split2, "fwd"), timed(split2, "bwd")), flush=True)
