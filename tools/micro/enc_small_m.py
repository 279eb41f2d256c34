#!/usr/bin/env python3
"""encoder_fwd_kernel's time per observation against the batch size (the learner's 10 k rows against the actor's 160 k): the in-tree
library, mapf_encoder_forward (u8 and bf16 input) and mapf_encoder_forward_save.  Run on the GPU."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mapf_rl_amd._lib import lib  # noqa: E402

vp = ctypes.c_void_p
w = (torch.randn(894976, device="cuda") * 0.03).to(torch.float16)
b = torch.zeros(912, dtype=torch.float32, device="cuda")
for M in (2048, 5120, 10240, 10326, 12288, 20480, 40960, 163840):
    obs8 = (torch.rand((M, 6, 9, 9), device="cuda") < 0.3).to(torch.uint8)
    obs16 = obs8.to(torch.bfloat16)
    lat = torch.empty((M, 784), dtype=torch.bfloat16, device="cuda")
    acts = torch.empty((7, M, 49, 128), dtype=torch.float16, device="cuda")
    bits = torch.empty((7, M, 49, 4), dtype=torch.int32, device="cuda")
    res = []
    for name, args in (("u8", (obs8.data_ptr(), 0)), ("bf16", (obs16.data_ptr(), 1))):
        for save in (False, True):
            fn = lib.mapf_encoder_forward_save if save else lib.mapf_encoder_forward
            a = [vp(args[0]), args[1], M, vp(w.data_ptr()), vp(b.data_ptr()), vp(lat.data_ptr())] + ([vp(acts.data_ptr()), vp(bits.data_ptr())] if save else []) + [None]
            for _ in range(3):
                rc = fn(*a)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 20 if M < 50000 else 5
            e0.record()
            for _ in range(n):
                fn(*a)
            e1.record()
            torch.cuda.synchronize()
            res.append("%s%s %.3f ms = %.1f ns/obs" % (name, "+save" if save else "", e0.elapsed_time(e1) / n, e0.elapsed_time(e1) / n * 1e6 / M))
    print("M=%6d  " % M + "   ".join(res), flush=True)
