"""Shared by tests/test_big_goldens_cpu.py (fp32 on CPU, 1e-4) and tests/test_big_goldens_gpu.py (bf16 kernels): access to
tests/golden/dqn_big.npz -- reference Network.step at 64 agents, and a replay-shaped batch of real observations at A = 40
(BASELINE shape) / A = 6 with the reference's bootstrap Q-values, one Learner.train body and per-parameter gradient
fingerprints (tests/golden/make_dqn_goldens_big.py)."""
import numpy as np
import torch

from tests import helpers as H


def batch(z, tag, device="cpu", obs_dtype=torch.float32):
    """The sample_batch 11-tuple of batch `tag` ('b40' / 'b6')."""
    pre = tag + "_"
    B, T, A, L = [int(v) for v in z[pre + "shape"]]
    obs = torch.from_numpy(H.unpack_bits(z[pre + "obs_bits"], (B, T, A, 6, 9, 9))).to(device, obs_dtype)
    comm = torch.from_numpy(H.unpack_bits(z[pre + "comm_bits"], (B, T, A, A)).astype(bool)).to(device)
    hidden = torch.from_numpy(np.repeat(z[pre + "hidden0"][:, None, :], A, axis=1).reshape(B * A, 256)).to(device)  # quirk Q4
    t = lambda k: torch.from_numpy(z[pre + k]).to(device)
    return (obs, t("action"), t("reward"), t("done"), t("steps"), t("bt_steps"), hidden, comm, None, t("weights"), 0)


def grad_errors(z, tag, named_grads, floor=1e-6):
    """{name: (estimated ||g - g_ref|| / ||g_ref||, worst per-gate block-norm relative error)} from the golden's fingerprints:
    the rms over the 24 +-1 projections of (g - g_ref) estimates the error's norm (exact for tensors stored in full).
    Norms below `floor` x the global gradient norm count as that floor (W_K.bias has an exactly zero gradient: a constant
    added to every key shifts all scores of a row alike)."""
    pre = tag + "_"
    out = {}
    fl = floor * float(z[pre + "grad_norm"])
    for name in [str(n) for n in z[pre + "grad_names"]]:
        g = np.asarray(named_grads[name], np.float64)
        ref_norm = float(z[pre + "gnorm_" + name])
        if pre + "gfull_" + name in z.files:
            err = float(np.linalg.norm(g - z[pre + "gfull_" + name].astype(np.float64)))
        else:
            fp = H.grad_fingerprint(g)
            err = float(np.sqrt(np.mean((fp["proj"] - z[pre + "gproj_" + name]) ** 2)))
        blk = H.grad_fingerprint(g)["blocks"]
        blk_ref = z[pre + "gblk_" + name]
        out[name] = (err / max(ref_norm, fl), float(np.max(np.abs(blk - blk_ref) / np.maximum(blk_ref, fl))))
    return out
