"""CPU (fp32): the product Network / Learner against reference goldens at the BASELINE agent counts
(tests/golden/dqn_big.npz): Network.step on the reference's 64-agent fixture, target-style bootstrap and one full
Learner.train body on B=8 x T=18 x A=40 windows of real observations with zero-padded tails and all-False comm rows
(what GlobalBuffer.sample_batch hands out, worker.py:118-162), and EVERY parameter gradient through its fingerprint.
Tolerance 1e-4 * max(1, |x|); the bf16 kernels are compared in tests/test_big_goldens_gpu.py."""
import numpy as np
import pytest
import torch

from tests import big_golden as BG
from tests import helpers as H
from tests.test_learner_cpu import _models
from tests.test_model_cpu import _close, _net


def test_step_64_agents_matches_reference():
    z = H.load_npz("dqn_big.npz")
    net, _ = _net()
    net.reset()
    N = 64
    for t in range(z["step64_q"].shape[0]):
        obs = H.unpack_bits(z["step64_obs_bits"][t], (N, 6, 9, 9)).astype(np.float32)
        cm = np.unpackbits(z["step64_comm_mask"][t], axis=-1, bitorder="little")[:, :N].astype(bool)
        actions, q, hidden, _ = net.step(obs, z["step64_pos"][t].astype(np.float32), comm_mask=cm)
        assert _close(q, z["step64_q"][t]), (t, np.abs(q - z["step64_q"][t]).max())
        assert _close(hidden, z["step64_hidden"][t]), t
        gap = np.sort(z["step64_q"][t], axis=1)
        clear = (gap[:, -1] - gap[:, -2]) > 1e-3
        assert np.array_equal(np.array(actions)[clear], z["step64_actions"][t][clear])


@pytest.mark.parametrize("tag", ["b40", "b6", "b128"])
def test_update_matches_reference_at_baseline_shape(tag):
    z = H.load_npz("dqn_big.npz")
    pre = tag + "_"
    lr = _models()
    b = BG.batch(z, tag)
    # windows contain rows whose comm mask is all False (zero padding, worker.py:139-142): they must not produce NaN
    assert not bool(b[7].any(-1).all())
    with torch.no_grad():
        nxt = b[5] + b[4].view(-1).long()
        q_tar = lr.tar_model.bootstrap(b[0], nxt, b[6], b[7])
    assert _close(q_tar.numpy(), z[pre + "q_target_all"]), np.abs(q_tar.numpy() - z[pre + "q_target_all"]).max()
    grads = {}
    orig_clip = torch.nn.utils.clip_grad_norm_

    def grab(params, max_norm):  # gradients as the reference sees them: after backward, before the clip (worker.py:316-319)
        for k, p in lr.model.named_parameters():
            grads[k] = p.grad.detach().clone().numpy()
        return orig_clip(lr.model.parameters(), max_norm)

    torch.nn.utils.clip_grad_norm_ = grab
    try:
        out = lr.update(b)
    finally:
        torch.nn.utils.clip_grad_norm_ = orig_clip
    assert _close(out["q_next"].numpy(), z[pre + "q_next"]) and _close(out["td"].numpy(), z[pre + "td"])
    assert np.allclose(out["priorities"].numpy(), z[pre + "priorities"], rtol=1e-4, atol=1e-6)
    assert abs(float(out["loss"]) - float(z[pre + "loss"])) <= 1e-5 * max(1, abs(float(z[pre + "loss"])))
    assert abs(float(out["grad_norm"]) - float(z[pre + "grad_norm"])) <= 1e-3 * float(z[pre + "grad_norm"])
    errs = BG.grad_errors(z, tag, grads)
    assert len(errs) == 35
    worst = max(errs.items(), key=lambda kv: kv[1][0])
    assert worst[1][0] <= 1e-3, worst          # fp32 vs fp32: every parameter's gradient, relative to its own norm
    assert max(v[1] for v in errs.values()) <= 1e-3
