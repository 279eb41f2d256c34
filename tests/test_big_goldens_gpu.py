"""GPU (16-bit MFMA kernels): the product path against REFERENCE goldens at the BASELINE agent counts (tests/golden/dqn_big.npz,
captured by tests/golden/make_dqn_goldens_big.py from the unmodified reference in fp32): Network.step on the reference's
64-agent fixture, and one Learner.train body on replay-shaped batches of real observations at A = 40 (config 2), A = 6 (the
reference's training shape) and A = 128 (config 5) -- Q-values, td error, loss, gradient norm and EVERY parameter's gradient
(through its fingerprint).  Tolerances: |dQ| <= 2e-2 max(1, |Q|) per bootstrap (td: two bootstraps), and per
parameter tensor ||g - g_ref|| <= BOUND ||g_ref|| (estimated from 24 fixed +-1 projections; exact for small tensors), with
gradient norms below 1e-3 of the global norm measured against that floor (bf16 rounding noise of the other tensors):
  * BOUND = GRAD_BOUND[tensor] below: 1.5 x the error measured for that tensor (round 5); the recurrence and the Q head (GRU cells,
    attention, W_O, adv/state: the BPTT kernels) measure 1e-3 .. 6e-3;
  * the 16 encoder tensors measure 0.007 .. 0.037: their gradient is a sum over ~10^6 positions through 8 layers of 16-bit activations
    with strong cancellation.  The encoder kernels keep activations, weights and pre-activation gradients in f16 (the reference's own
    AMP format, worker.py:283) and measure 0.009-0.037 against this golden; in bf16 (rounds 1-2; and the layer-by-layer MIOpen path
    under bf16 autocast today) the same tensors sit 0.06-0.14 off -- the 3 mantissa bits of the ACTIVATIONS are what matters, the
    format of the gradients does not (profiles/r03_encoder_grad_error_fp16.txt); the same module path in fp32 on the GPU measures
    1e-6 (tools/enc_grad_check.py); every tensor's NORM (and every GRU gate block's) agrees within 2e-2."""
import numpy as np
import pytest
import torch

from tests import big_golden as BG
from tests import helpers as H
from tests.test_learner_cpu import _models

pytestmark = pytest.mark.gpu


# Per-tensor bounds on ||g - g_ref|| / ||g_ref|| against the reference's fp32 gradients (worker.py:316-323): 1.5 x the largest error
# measured over the three goldens (A = 40 / 6 / 128; profiles/r05_grad_errors_vs_reference.txt; floor 2e-3), so every tensor has its own
# margin instead of a flat 0.04 / 0.02 (round-4 review: the flat 0.04 left the worst encoder tensors no room, and hid that the 1x1 head
# and the whole recurrence sit at a third of it).  The kernels are deterministic: these numbers reproduce run to run.
GRAD_BOUND = {
    "obs_encoder.0.weight":              0.042,   # measured 2.8e-02 / 1.5e-02 / 7.8e-03
    "obs_encoder.0.bias":                0.054,   # measured 3.6e-02 / 1.9e-02 / 8.6e-03
    "obs_encoder.2.block1.weight":       0.045,   # measured 3.0e-02 / 2.3e-02 / 1.9e-02
    "obs_encoder.2.block1.bias":         0.047,   # measured 3.1e-02 / 2.1e-02 / 1.4e-02
    "obs_encoder.2.block2.weight":       0.042,   # measured 2.8e-02 / 2.3e-02 / 8.1e-03
    "obs_encoder.2.block2.bias":         0.047,   # measured 3.1e-02 / 2.0e-02 / 8.3e-03
    "obs_encoder.3.block1.weight":       0.036,   # measured 2.4e-02 / 2.3e-02 / 1.2e-02
    "obs_encoder.3.block1.bias":         0.051,   # measured 3.4e-02 / 2.5e-02 / 9.0e-03
    "obs_encoder.3.block2.weight":       0.048,   # measured 3.2e-02 / 1.5e-02 / 8.1e-03
    "obs_encoder.3.block2.bias":         0.045,   # measured 3.0e-02 / 1.9e-02 / 8.1e-03
    "obs_encoder.4.block1.weight":       0.056,   # measured 3.7e-02 / 2.7e-02 / 1.2e-02
    "obs_encoder.4.block1.bias":         0.051,   # measured 3.4e-02 / 2.9e-02 / 1.2e-02
    "obs_encoder.4.block2.weight":       0.03,   # measured 2.0e-02 / 6.8e-03 / 7.2e-03
    "obs_encoder.4.block2.bias":         0.042,   # measured 2.8e-02 / 8.6e-03 / 6.8e-03
    "obs_encoder.5.weight":              0.013,   # measured 8.1e-03 / 5.5e-03 / 7.4e-03
    "obs_encoder.5.bias":                0.014,   # measured 8.9e-03 / 5.1e-03 / 7.7e-03
    "recurrent.weight_ih":               0.0086,   # measured 5.7e-03 / 4.1e-03 / 4.2e-03
    "recurrent.weight_hh":               0.0086,   # measured 4.5e-03 / 5.7e-03 / 4.4e-03
    "recurrent.bias_ih":                 0.0077,   # measured 5.0e-03 / 3.9e-03 / 5.1e-03
    "recurrent.bias_hh":                 0.0081,   # measured 5.1e-03 / 3.9e-03 / 5.4e-03
    "comm.self_attn.W_Q.weight":         0.0029,   # measured 1.5e-03 / 6.0e-04 / 1.9e-03
    "comm.self_attn.W_Q.bias":           0.0027,   # measured 1.0e-03 / 5.2e-04 / 1.8e-03
    "comm.self_attn.W_K.weight":         0.0035,   # measured 1.5e-03 / 5.0e-04 / 2.3e-03
    "comm.self_attn.W_K.bias":           0.0027,   # measured 8.6e-04 / 3.5e-04 / 1.8e-03
    "comm.self_attn.W_V.weight":         0.0062,   # measured 3.7e-03 / 3.4e-03 / 4.1e-03
    "comm.self_attn.W_V.bias":           0.0053,   # measured 3.3e-03 / 3.3e-03 / 3.5e-03
    "comm.self_attn.W_O.weight":         0.0068,   # measured 4.5e-03 / 3.3e-03 / 4.3e-03
    "comm.update_cell.weight_ih":        0.0089,   # measured 5.9e-03 / 4.4e-03 / 4.7e-03
    "comm.update_cell.weight_hh":        0.0069,   # measured 3.0e-03 / 4.6e-03 / 3.8e-03
    "comm.update_cell.bias_ih":          0.0045,   # measured 2.9e-03 / 2.5e-03 / 3.0e-03
    "comm.update_cell.bias_hh":          0.0048,   # measured 3.0e-03 / 3.0e-03 / 3.2e-03
    "adv.weight":                        0.0081,   # measured 3.7e-03 / 5.4e-03 / 3.4e-03
    "adv.bias":                          0.002,   # measured 3.2e-04 / 1.3e-03 / 7.8e-04
    "state.weight":                      0.0068,   # measured 4.0e-03 / 4.5e-03 / 3.3e-03
    "state.bias":                        0.027,   # measured 4.9e-04 / 1.8e-02 / 5.0e-04
}
del GRAD_BOUND["state.bias"], GRAD_BOUND["adv.bias"]  # checked against their closed form below, not against a measured number


def _close(a, b, tol):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.all(np.abs(a - b) <= tol * np.maximum(1.0, np.abs(b)))


def test_step_64_agents_fused_kernels_vs_reference():
    """The reference's own 64-agent fixture (test64_40_0.3.pkl, test.py:82-145) through the fused encoder + the wide
    recurrence kernel (csrc/mapf_recur_wide.hip)."""
    from mapf_rl_amd.model import Network

    z = H.load_npz("dqn_big.npz")
    net = Network().cuda().eval()
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    net.load_state_dict({k: torch.from_numpy(v) for k, v in H.det_state_dict(shapes, seed=1234).items()})
    assert Network.FUSED_RECURRENCE and Network.FUSED_INFERENCE
    N = 64
    net.reset()
    for t in range(z["step64_q"].shape[0]):
        obs = torch.from_numpy(H.unpack_bits(z["step64_obs_bits"][t], (N, 6, 9, 9))).cuda()
        cm = np.unpackbits(z["step64_comm_mask"][t], axis=-1, bitorder="little")[:, :N].astype(bool)
        actions, q, hidden, _ = net.step(obs, torch.from_numpy(z["step64_pos"][t].astype(np.int16)).cuda(), comm_mask=cm)
        assert _close(q, z["step64_q"][t], 2e-2), (t, np.abs(q - z["step64_q"][t]).max())
        assert _close(hidden, z["step64_hidden"][t], 2e-2), t
        gap = np.sort(z["step64_q"][t], axis=1)
        clear = (gap[:, -1] - gap[:, -2]) > 4e-2
        assert np.array_equal(np.array(actions)[clear], z["step64_actions"][t][clear])


@pytest.mark.parametrize("tag", ["b40", "b6", "b128"])
def test_update_bf16_kernels_vs_reference(tag):
    z = H.load_npz("dqn_big.npz")
    pre = tag + "_"
    lr = _models("cuda")
    b = BG.batch(z, tag, "cuda", torch.bfloat16)
    with torch.no_grad():
        nxt = b[5] + b[4].view(-1).long()
        q_tar = lr.tar_model.bootstrap(b[0], nxt, b[6], b[7])
    assert _close(q_tar.cpu().numpy(), z[pre + "q_target_all"], 2e-2), np.abs(q_tar.cpu().numpy() - z[pre + "q_target_all"]).max()
    grads = {}

    def grab(learner):  # gradients as the reference sees them: after backward, before the clip (worker.py:316-319)
        for k, p in learner.model.named_parameters():
            grads[k] = p.grad.detach().float().cpu().numpy()

    lr.grad_hook = grab
    out = lr.update(b)
    td, ref = out["td"].float().cpu().numpy(), z[pre + "td"]
    assert np.all(np.isfinite(td))
    assert _close(td, ref, 4e-2), np.abs(td - ref).max()  # 2 bootstraps: 2 x 2e-2
    assert abs(float(out["loss"]) - float(z[pre + "loss"])) <= 5e-2 * max(1.0, float(z[pre + "loss"]))
    assert abs(float(out["grad_norm"]) - float(z[pre + "grad_norm"])) <= 2e-2 * float(z[pre + "grad_norm"])
    errs = BG.grad_errors(z, tag, grads, floor=1e-3)
    print("\nGRADERR %s %s" % (tag, {k: "%.1e/%.1e" % v for k, v in errs.items()}))  # (the measured error of every tensor, in the test log)
    assert len(errs) == 35
    for name, (err, blk) in errs.items():
        if name in ("state.bias", "adv.bias"):
            continue
        assert err <= GRAD_BOUND[name], (name, err, GRAD_BOUND[name])
        assert blk <= 2e-2, (name, blk)
    # state.bias / adv.bias: 1 / 5 numbers that are plain weighted sums of the clipped td errors over the batch -- with
    # loss = mean(w * huber(td)) (worker.py:310, :341-344) and q = V + A - mean(A) (model.py:259-262):
    #     d loss / d state.bias  = mean_b w_b clip(td_b, -1, 1)
    #     d loss / d adv.bias[a] = mean_b w_b clip(td_b, -1, 1) (1[a = a_b] - 1/5)
    # They are held to that closed form evaluated on the update's OWN td errors (a kernel check: 2e-3 of the vector's norm), and
    # the td errors to the reference's within the td tolerance above -- instead of round 4's flat 6e-2 against the reference's
    # gradient, which mixed the two (bf16 Q-values move small td errors by a few per cent).
    w = b[9].float().view(-1).cpu().numpy().astype(np.float64)
    act = b[1].view(-1).cpu().numpy()
    g = w * np.clip(td.reshape(-1).astype(np.float64), -1.0, 1.0) / td.size
    want_state = np.array([g.sum()])
    want_adv = np.array([(g * ((act == a) - 0.2)).sum() for a in range(5)])
    for name, want in (("state.bias", want_state), ("adv.bias", want_adv)):
        got = grads[name].astype(np.float64).reshape(-1)
        assert np.linalg.norm(got - want) <= 2e-3 * np.linalg.norm(want) + 1e-7, (name, got, want)
