#!/usr/bin/env python3
"""Which library kernel runs the input projection at a given row count (run under rocprofv3 --kernel-trace --stats)."""
import sys

import torch

w = (torch.randn(768, 784, device="cuda") * 0.05).to(torch.bfloat16)
for rows in [int(a) for a in sys.argv[1:]]:
    x = (torch.randn(rows, 784, device="cuda") * 0.5).to(torch.bfloat16)
    for _ in range(3):
        torch.mm(x, w.t())
    torch.cuda.synchronize()
