#!/usr/bin/env python3
"""Runs the three hand-written MFMA kernels of the encoder once each at the learner's shape (122,880 observations:
training forward with saved activations, backward-data chain, six weight-gradient launches) and the inference
forward at the actor's shape (163,840) -- the command behind the rocprofv3 kernel-trace / PMC summaries."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mapf_rl_amd.model import Network  # noqa: E402

torch.manual_seed(0)
net = Network().cuda()
reps = int(os.environ.get("REPS", 2))
obs_a = (torch.rand((163840, 6, 9, 9), device="cuda") < 0.3).to(torch.uint8)
obs_l = (torch.rand((122880, 6, 9, 9), device="cuda") < 0.3).to(torch.bfloat16)
for _ in range(reps):
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        net.encode(obs_a)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        lat = net.encode(obs_l)
    (lat.float() ** 2).mean().backward()
torch.cuda.synchronize()
print("ok")
