#!/usr/bin/env python3
"""Where does the encoder's gradient error against the fp32 reference golden come from?  One Learner.update on the b40 golden
batch with (a) the fused encoder training kernels, (b) the layer-by-layer MIOpen path under the same bf16 autocast, (c) the same
module path in fp32 on the GPU; per-parameter error estimate against the reference fingerprints (tests/big_golden.py)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from tests import big_golden as BG, helpers as H
from tests.test_learner_cpu import _models
from mapf_rl_amd.model import Network

z = H.load_npz("dqn_big.npz")
tag = sys.argv[1] if len(sys.argv) > 1 else "b40"


def run(fused_training, fp32=False):
    from mapf_rl_amd.learner import Learner

    Network.FUSED_TRAINING = fused_training
    Learner.FUSED_UPDATE = fused_training and not fp32  # (a): update.FusedUpdate; (b), (c): autograd over Network.bootstrap
    try:
        lr = _models("cuda")
        b = BG.batch(z, tag, "cuda", torch.float32 if fp32 else torch.bfloat16)
        grads = {}
        lr.grad_hook = lambda l: grads.update({k: p.grad.detach().float().cpu().numpy() for k, p in l.model.named_parameters()})
        if fp32:
            lr.model._autocast = lambda dev: torch.autocast("cuda", enabled=False)
            lr.tar_model._autocast = lr.model._autocast
        out = lr.update(b)
    finally:
        Network.FUSED_TRAINING = True
        Learner.FUSED_UPDATE = True
    return BG.grad_errors(z, tag, grads, floor=1e-3), grads, out

ea, ga, oa = run(True)
eb, gb, ob = run(False)
ec, gc, oc = run(False, fp32=True)
print("%-34s %10s %10s %10s %12s" % ("parameter", "fused", "miopen-bf16", "gpu-fp32", "fused-vs-miopen"))
for k in ea:
    d = np.linalg.norm(ga[k] - gb[k]) / max(np.linalg.norm(gb[k]), 1e-12)
    print("%-34s %10.2e %10.2e %10.2e %12.2e" % (k, ea[k][0], eb[k][0], ec[k][0], d))
print("td fused", oa["td"].float().cpu().numpy().ravel().round(4))
print("td miopn", ob["td"].float().cpu().numpy().ravel().round(4))
print("td fp32 ", oc["td"].float().cpu().numpy().ravel().round(4))
print("td ref  ", z[tag + "_td"].ravel().round(4))
