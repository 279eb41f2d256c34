#!/usr/bin/env python3
"""What would 16-bit floating point with 3 more mantissa bits buy the encoder's gradients?  

The fused encoder kernels keep layer outputs (activations) and pre-activation gradients (gz) in bf16; their weight gradients sit
6-11 % off the fp32 reference direction (tests/test_big_goldens_gpu.py holds them to 0.15).  The reference itself trains under fp16
autocast + GradScaler (worker.py:283,316-323).  Before rebuilding four MFMA kernels for f16, this emulation separates the two rounding
points: the encoder (reference model.py:147-162) in fp32 PyTorch with the activations rounded to a 16-bit format at every layer
output -- where the kernels round -- and the gradient rounded at every pre-activation, both independently bf16 / f16 (static loss
scale) / fp32.  Input: the observations of the reference-captured batch b40 (tests/golden/dqn_big.npz) and the gradient w.r.t. the
encoder's output taken from the same update run in fp32.  Output: per variant the error of every encoder parameter's gradient
against the fp32 gradient, ||g - g32|| / ||g32||."""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mapf_rl_amd.learner import Learner  # noqa: E402
from mapf_rl_amd.model import Network  # noqa: E402
from tests import big_golden as BG, helpers as H  # noqa: E402
from tests.test_learner_cpu import _models  # noqa: E402

z = H.load_npz("dqn_big.npz")
tag = sys.argv[1] if len(sys.argv) > 1 else "b40"
SCALE = float(os.environ.get("LOSS_SCALE", 4096.0))

# ---- the fp32 update: encoder input, gradient w.r.t. the encoder's output, fp32 parameter gradients ----
Learner.FUSED_UPDATE = False
Network.FUSED_TRAINING = Network.FUSED_INFERENCE = Network.FUSED_EPILOGUE = False
lr = _models("cuda")
lr.model._autocast = lambda dev: torch.autocast("cuda", enabled=False)
lr.tar_model._autocast = lr.model._autocast
grab = {}
orig_encode = lr.model.encode


def encode_hook(obs):
    out = orig_encode(obs)
    if torch.is_grad_enabled():
        grab["obs"] = obs.detach().float()
        out.register_hook(lambda g: grab.__setitem__("g_lat", g.detach().float()))
    return out


lr.model.encode = encode_hook
enc_params0 = {k: p.detach().clone() for k, p in lr.model.obs_encoder.named_parameters()}  # before the optimizer step
g32 = {}
lr.grad_hook = lambda l: g32.update({k: p.grad.detach().float().clone() for k, p in l.model.named_parameters() if k.startswith("obs_encoder.")})
lr.update(BG.batch(z, tag, "cuda", torch.float32))
obs, g_lat = grab["obs"], grab["g_lat"]
print("%s: %d observations through the encoder (pruned update), |g_lat| max %.3e, median nonzero %.3e" % (
    tag, obs.shape[0], float(g_lat.abs().max()), float(g_lat[g_lat != 0].abs().median())))


class Round(torch.autograd.Function):
    """forward: round to `fwd` (None: keep); backward: round the gradient to `bwd` with a static scale (None: keep)."""

    @staticmethod
    def forward(ctx, x, fwd, bwd, scale):
        ctx.bwd, ctx.scale = bwd, scale
        return x if fwd is None else x.to(fwd).float()

    @staticmethod
    def backward(ctx, g):
        if ctx.bwd is not None:
            g = (g * ctx.scale).to(ctx.bwd).float() / ctx.scale
        return g, None, None, None


def encoder_grads(act_dt, gz_dt, weight_dt):
    params = {k: p.clone().requires_grad_(True) for k, p in enc_params0.items()}
    w = lambda k: params[k] if weight_dt is None else Round.apply(params[k], weight_dt, None, 1.0)  # weights as the MFMA kernels see them
    pre = lambda t: Round.apply(t, None, gz_dt, SCALE)   # gradient w.r.t. a pre-activation: where the backward kernel rounds
    post = lambda t: Round.apply(t, act_dt, None, 1.0)   # a layer's output: where the forward kernel rounds
    x = obs
    h = post(F.relu(pre(F.conv2d(x, w("0.weight"), params["0.bias"]))))
    for b in ("2", "3", "4"):
        t = post(F.relu(pre(F.conv2d(h, w(b + ".block1.weight"), params[b + ".block1.bias"], 1, 1))))
        h = post(F.relu(pre(F.conv2d(t, w(b + ".block2.weight"), params[b + ".block2.bias"], 1, 1) + h)))
    out = post(F.relu(pre(F.conv2d(h, w("5.weight"), params["5.bias"])))).flatten(1)
    g = g_lat if gz_dt is None else (g_lat * SCALE).to(gz_dt).float() / SCALE  # the incoming gradient arrives in that format too
    out.backward(g)
    return {"obs_encoder." + k: p.grad for k, p in params.items()}


variants = [("act bf16, gz bf16 (the kernels today)", torch.bfloat16, torch.bfloat16, torch.bfloat16),
            ("act bf16, gz f16 x%g (VERDICT's variant)" % SCALE, torch.bfloat16, torch.float16, torch.bfloat16),
            ("act bf16, gz fp32 (bound of fixing gz only)", torch.bfloat16, None, torch.bfloat16),
            ("act f16,  gz f16 x%g (all f16: the reference's AMP)" % SCALE, torch.float16, torch.float16, torch.float16),
            ("act f16,  gz bf16", torch.float16, torch.bfloat16, torch.float16),
            ("act fp32, gz bf16 (bound of fixing activations only)", None, torch.bfloat16, None),
            ("act fp32, gz fp32 (sanity: must be ~0)", None, None, None)]
names = sorted(g32)
rows = []
for label, a, g, wd in variants:
    gr = encoder_grads(a, g, wd)
    errs = [float((gr[k] - g32[k]).norm() / g32[k].norm()) for k in names]
    rows.append((label, errs))
    print("%-52s weights: mean %.3f max %.3f   biases: mean %.3f max %.3f" % (
        label, np.mean([e for k, e in zip(names, errs) if k.endswith("weight")]), np.max([e for k, e in zip(names, errs) if k.endswith("weight")]),
        np.mean([e for k, e in zip(names, errs) if k.endswith("bias")]), np.max([e for k, e in zip(names, errs) if k.endswith("bias")])), flush=True)
print("\nper tensor:")
print("%-34s " % "parameter" + " ".join("%9s" % ("v%d" % i) for i in range(len(variants))))
for j, k in enumerate(names):
    print("%-34s " % k + " ".join("%9.2e" % rows[i][1][j] for i in range(len(variants))))
