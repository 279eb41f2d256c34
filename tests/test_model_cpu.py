"""CPU (fp32): the product Network against golden vectors from the reference Network (model.py) with
deterministic weights.  Tolerance |dQ| <= 1e-4 * max(1, |Q|); GPU/bf16 parity is in
tests/test_model_gpu.py."""
import numpy as np
import pytest
import torch

from tests import helpers as H


def _net():
    from mapf_rl_amd.model import Network

    net = Network()
    net.eval()
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    sd = H.det_state_dict(shapes, seed=1234)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return net, shapes


def _close(a, b, tol=1e-4):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.all(np.abs(a - b) <= tol * np.maximum(1.0, np.abs(b)))


def test_state_dict_names_and_size_match_reference():
    z = H.load_npz("dqn_model.npz")
    net, shapes = _net()
    assert list(shapes.keys()) == [str(n) for n in z["param_names"]]
    assert [int(np.prod(s)) for s in shapes.values()] == z["param_numel"].tolist()
    assert sum(int(np.prod(s)) for s in shapes.values()) == 2050582


@pytest.mark.parametrize("nag", [16, 32])
def test_step_matches_reference_with_injected_mask(nag):
    z = H.load_npz("dqn_model.npz")
    net, _ = _net()
    pre = "step%d_" % nag
    T = z[pre + "q"].shape[0]
    net.reset()
    for t in range(T):
        obs = H.unpack_bits(z[pre + "obs_bits"][t], (nag, 6, 9, 9)).astype(np.float32)
        pos = z[pre + "pos"][t].astype(np.float32)
        actions, q, hidden, cm = net.step(obs, pos, comm_mask=z[pre + "comm_mask"][t])
        assert _close(q, z[pre + "q"][t]), (nag, t, np.abs(q - z[pre + "q"][t]).max())
        assert _close(hidden, z[pre + "hidden"][t]), (nag, t)
        gap = np.sort(z[pre + "q"][t], axis=1)
        clear = (gap[:, -1] - gap[:, -2]) > 1e-3
        assert np.array_equal(np.array(actions)[clear], z[pre + "actions"][t][clear])


@pytest.mark.parametrize("nag", [16, 32])
def test_comm_mask_selection(nag):
    """Own 3-nearest selection: exact on rows without a distance tie at the cut, a valid top-k set otherwise."""
    from mapf_rl_amd.model import comm_mask_from_pos

    z = H.load_npz("dqn_model.npz")
    pre = "step%d_" % nag
    for t in range(z[pre + "q"].shape[0]):
        pos = z[pre + "pos"][t].astype(np.int64)
        ours = comm_mask_from_pos(torch.from_numpy(pos)[None])[0].numpy()
        ref = z[pre + "comm_mask"][t]
        d = pos[:, None, :] - pos[None, :, :]
        d2 = (d ** 2).sum(-1)
        in_fov = (np.abs(d) <= 4).all(-1)
        srt = np.sort(d2, axis=1)
        tie_free = srt[:, 2] != srt[:, 3]
        assert np.array_equal(ours[tie_free], ref[tie_free])
        assert np.all(ours.diagonal())
        # tied rows: a valid answer = in FOV and among agents at distance <= the 3rd smallest
        ok = in_fov & (d2 <= srt[:, 2:3])
        assert np.all(~ours | ok)
        assert np.all(ours.sum(1) <= 3)


def test_bootstrap_matches_reference():
    z = H.load_npz("dqn_model.npz")
    net, _ = _net()
    B, T, A = z["boot_shape"]
    obs = H.unpack_bits(z["boot_obs_bits"], (B, T, A, 6, 9, 9)).astype(np.float32)
    with torch.no_grad():
        q = net.bootstrap(torch.from_numpy(obs), torch.from_numpy(z["boot_steps"]), torch.from_numpy(z["boot_hidden"]),
                          torch.from_numpy(z["boot_comm"]))
    assert q.shape == (B, 5)
    assert _close(q.numpy(), z["boot_q"]), np.abs(q.numpy() - z["boot_q"]).max()


def test_step_batch_equals_per_env_steps():
    """E environments in one batched call == E separate reference-style calls (no cross-env leakage)."""
    net, _ = _net()
    rng = np.random.RandomState(0)
    E, N = 3, 5
    obs = torch.from_numpy((rng.random_sample((E, N, 6, 9, 9)) < 0.3).astype(np.float32))
    pos = torch.from_numpy(rng.randint(0, 12, size=(E, N, 2)))
    a, q, h, cm = net.step_batch(obs, pos, None)
    a2, q2, h2, cm2 = net.step_batch(obs, pos, h)
    for e in range(E):
        net.reset()
        ae, qe, he, cme = net.step(obs[e], pos[e])
        assert _close(qe, q[e].numpy()) and np.array_equal(cme, cm[e].numpy())
        ae, qe, he, cme = net.step(obs[e], pos[e])
        assert _close(qe, q2[e].numpy(), 1e-4)


def test_deferred_weight_gradients_match_autograd():
    """`_TimeLinear` + `_WGradSink` (mapf_rl_amd/model.py): a T-step recurrence whose per-step weight gradients are
    deferred to one GEMM per weight at the end of backward gives the same gradients as plain autograd -- including
    the first step (input without grad), a fused two-parameter key with row slices, and gradient accumulation
    into an existing .grad."""
    import torch
    import torch.nn.functional as F

    from mapf_rl_amd.model import _TimeLinear, _WGradSink

    torch.manual_seed(0)
    T, R, D = 5, 7, 6
    lin_a, lin_b = torch.nn.Linear(D, D), torch.nn.Linear(D, D)
    lin_o = torch.nn.Linear(2 * D, D, bias=False)
    x0 = torch.randn(R, D)
    target = torch.randn(R, D)

    def run(deferred):
        for p in list(lin_a.parameters()) + list(lin_b.parameters()) + list(lin_o.parameters()):
            p.grad = torch.ones_like(p)                      # accumulation into an existing gradient
        sink = _WGradSink()
        w_ab = torch.cat([lin_a.weight, lin_b.weight]).detach()
        b_ab = torch.cat([lin_a.bias, lin_b.bias]).detach()
        key_ab = [(lin_a.weight, lin_a.bias, 0, D), (lin_b.weight, lin_b.bias, D, 2 * D)]
        key_o = [(lin_o.weight, None, 0, D)]
        h = x0
        for _ in range(T):
            if deferred:
                ab = _TimeLinear.apply(h, w_ab, b_ab, sink, key_ab, lin_a.weight)
                h = torch.tanh(_TimeLinear.apply(torch.relu(ab), lin_o.weight.detach(), None, sink, key_o, lin_o.weight))
            else:
                ab = torch.cat([lin_a(h), lin_b(h)], dim=-1)
                h = torch.tanh(lin_o(torch.relu(ab)))
        ((h - target) ** 2).sum().backward()
        assert not sink.items and not sink.queued            # flushed by the end-of-backward callback
        return [p.grad.clone() for p in (lin_a.weight, lin_a.bias, lin_b.weight, lin_b.bias, lin_o.weight)]

    ref, got = run(False), run(True)
    for a, b in zip(ref, got):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-6)
        assert float((a - 1).abs().max()) > 1e-3             # a real gradient was added to the ones


def test_fragment_packing_layouts():
    """The MFMA A-fragment order documented in include/mapf_dqn.h for mapf_recurrent_infer's weights: element (o, k) of an
    [O, K] matrix sits at tile o/16, k-step k/32, lane 16*((k%32)//8) + o%16, slot k%8 (pure index check, no GPU)."""
    import torch

    O, K = 48, 96
    m = torch.arange(O * K, dtype=torch.float32).reshape(O, K)
    packed = m.reshape(O // 16, 16, K // 32, 4, 8).permute(0, 2, 3, 1, 4).reshape(-1)   # the expression in fused.PackedRecurrence
    for o, k in [(0, 0), (5, 7), (17, 40), (47, 95), (16, 32), (31, 63)]:
        tile, ks, lane, slot = o // 16, k // 32, 16 * ((k % 32) // 8) + o % 16, k % 8
        assert float(packed[((tile * (K // 32) + ks) * 64 + lane) * 8 + slot]) == float(m[o, k])
