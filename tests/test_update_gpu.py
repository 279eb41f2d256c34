"""GPU: the kernels of csrc/mapf_update.hip (plan, dueling head + loss, weight packing, bias gradients, clip + Adam) against plain
PyTorch statements of the same reference code (worker.py:296-324,341-344; model.py:242-262), and `update.FusedUpdate` -- the explicit
forward / backward `Learner.update` runs on a HIP device -- against the autograd path over `Network.bootstrap` on the
reference-captured batches."""
import ctypes

import numpy as np
import pytest
import torch

from tests import big_golden as BG
from tests import helpers as H
from tests.test_learner_cpu import _batch, _models
from tests.test_relevance_gpu import _torch_relevance

pytestmark = pytest.mark.gpu


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _plan_mark(comm, steps, extra=None, mark_all=0, want_rel=True):
    from mapf_rl_amd._lib import check, lib

    B, T, N, _ = comm.shape
    cm = comm.view(torch.uint8)
    rel = torch.empty((T, B, N), dtype=torch.uint8, device="cuda") if want_rel else None
    slot = torch.empty((B, N), dtype=torch.int16, device="cuda")
    order = torch.empty((B, N), dtype=torch.int16, device="cuda")
    nact = torch.empty((T, B), dtype=torch.int32, device="cuda")
    cnt = torch.empty(B, dtype=torch.int32, device="cuda")
    nag = torch.empty(B, dtype=torch.int32, device="cuda")
    check(lib.mapf_plan_mark(_p(cm), cm.stride(0), cm.stride(1), _p(steps), _p(extra), T, B, N, mark_all, _p(rel), _p(slot), _p(order), _p(nact),
                             _p(cnt), _p(nag), None, None), "mapf_plan_mark")
    torch.cuda.synchronize()
    return rel, slot, order, nact, cnt, nag


def _random_windows(B, T, N, p, seed, time_major=False):
    g = torch.Generator(device="cuda").manual_seed(seed)
    shape = (T, B, N, N) if time_major else (B, T, N, N)
    comm = (torch.rand(shape, device="cuda", generator=g) < p) | torch.eye(N, dtype=torch.bool, device="cuda")
    if time_major:
        comm = comm.transpose(0, 1)  # [B, T, N, N] view of time-major memory: the device replay's layout
    steps = torch.randint(1, T + 1, (B,), device="cuda", generator=g)
    return comm, steps, g


@pytest.mark.parametrize("B,T,N,p,tm", [(192, 16, 40, 0.04, True), (7, 18, 128, 0.01, False), (33, 18, 6, 0.2, True), (5, 3, 1, 1.0, False),
                                        (64, 18, 49, 0.0, False), (16, 20, 70, 0.02, True)])
def test_plan_mark_is_the_closure_in_prefix_order(B, T, N, p, tm):
    comm, steps, _ = _random_windows(B, T, N, p, B + N, tm)
    rel, slot, order, nact, cnt, nag = _plan_mark(comm, steps)
    want = _torch_relevance(comm, steps)
    assert torch.equal(rel.bool(), want)
    w = want.cpu().numpy()  # [T, B, N]
    slot, order, nact, cnt, nag = [x.cpu().numpy() for x in (slot, order, nact, cnt, nag)]
    assert np.array_equal(cnt, w.sum(axis=(0, 2))) and np.array_equal(nag, w.any(axis=0).sum(axis=1)) and np.array_equal(nact, w.sum(axis=2))
    for b in range(B):
        n = nag[b]
        assert order[b, 0] == 0 and slot[b, 0] == 0  # agent 0 stays agent 0
        assert sorted(order[b, :n].tolist()) == np.nonzero(w[:, b].any(axis=0))[0].tolist() and (order[b, n:] == -1).all()
        assert all(slot[b, order[b, i]] == i for i in range(n)) and (slot[b][~w[:, b].any(axis=0)] == -1).all()
        for t in range(T):  # the agents needed at step t are a prefix of the order
            assert set(order[b, :nact[t, b]].tolist()) == set(np.nonzero(w[t, b])[0].tolist())


def test_plan_mark_extra_steps_and_mark_all():
    B, T, N = 24, 18, 40
    comm, bt, g = _random_windows(B, T, N, 0.05, 3)
    bt = torch.clamp(bt, max=T - 2)
    extra = torch.randint(1, 3, (B,), device="cuda", generator=g).float()
    rel = _plan_mark(comm, bt, extra)[0]
    assert torch.equal(rel.bool(), _torch_relevance(comm, bt + extra.long()))
    rel, slot, order, nact, cnt, nag = _plan_mark(comm, bt, None, mark_all=1)
    last = (bt - 1).cpu().numpy()
    w = rel.cpu().numpy().astype(bool)
    assert all(w[:last[b] + 1, b].all() and not w[last[b] + 1:, b].any() for b in range(B))
    assert np.array_equal(order.cpu().numpy(), np.tile(np.arange(N), (B, 1))) and (nag.cpu().numpy() == N).all()


@pytest.mark.parametrize("B,T,N,p,tm", [(48, 16, 40, 0.04, True), (5, 18, 128, 0.01, False), (9, 18, 6, 0.3, True)])
def test_plan_rows_compact_layout(B, T, N, p, tm):
    from mapf_rl_amd._lib import check, lib

    comm, steps, g = _random_windows(B, T, N, p, 11 + N, tm)
    rel, slot, order, nact, cnt, nag = _plan_mark(comm, steps)
    obs_shape = (T, B, N, 486) if tm else (B, T, N, 486)
    obs = torch.rand(obs_shape, device="cuda", generator=g).to(torch.bfloat16)  # distinct rows
    if tm:
        obs = obs.transpose(0, 1)
    hidden = (torch.randn((B * N, 256), device="cuda", generator=g) * 0.3).to(torch.float16)
    Nc = 16 * -(-int(nag.max()) // 16)
    rows = int(cnt.sum())
    gidx = torch.empty((T, B, Nc), dtype=torch.int32, device="cuda")
    comm_c = torch.empty((T, B, Nc, Nc), dtype=torch.uint8, device="cuda")
    h0_c = torch.empty((B, Nc, 256), dtype=torch.bfloat16, device="cuda")
    obs_rows = torch.empty((rows, 486), dtype=torch.bfloat16, device="cuda")
    row_src = torch.empty(rows, dtype=torch.int64, device="cuda")
    cm = comm.view(torch.uint8)
    check(lib.mapf_plan_rows(T, B, N, Nc, _p(order), _p(nact), _p(cnt), _p(nag), _p(cm), cm.stride(0), cm.stride(1), _p(hidden), 0, _p(obs),
                             obs.stride(0), obs.stride(1), _p(gidx), _p(comm_c), _p(h0_c), rows, _p(row_src), _p(obs_rows), None, None, None, None, None), "mapf_plan_rows")
    torch.cuda.synchronize()
    G, O, NA = gidx.cpu().numpy(), order.cpu().numpy().astype(np.int64), nact.cpu().numpy()
    flat = [G[t, b, i] for b in range(B) for t in range(T) for i in range(NA[t, b])]
    assert flat == list(range(rows))  # window by window, step by step, position by position
    assert (G >= 0).sum() == rows and all((G[t, b, NA[t, b]:] == -1).all() for t in range(T) for b in range(B))
    ob, orows, cc, cmn = obs.float().cpu().numpy(), obs_rows.float().cpu().numpy(), comm_c.cpu().numpy(), comm.cpu().numpy()
    h0, hid = h0_c.float().cpu().numpy(), hidden.to(torch.bfloat16).float().cpu().numpy().reshape(B, N, 256)
    for b in range(0, B, max(1, B // 7)):
        n = int(nag[b])
        assert np.array_equal(h0[b, :n], hid[b, O[b, :n]]) and not h0[b, n:].any()
        for t in range(T):
            na = NA[t, b]
            for i in range(na):
                assert np.array_equal(orows[G[t, b, i]], ob[b, t, O[b, i]])
            want = np.eye(Nc, dtype=np.uint8)
            want[:na, :na] = cmn[b, t][np.ix_(O[b, :na], O[b, :na])]
            assert np.array_equal(cc[t, b], want)


def test_plan_rows_padded_initialises_the_padding_and_zero_rows_from_reads_its_bound_on_the_device():
    """mapf_plan_rows_padded (bucket-sized launches of the graph-replayed update): the real rows equal mapf_plan_rows', every padding row
    gathers observation 0 of the batch (row_src = 0) and -- with duplicate flags -- uses distinct row 0 and is skipped by the gradient
    sum (umap = 0, row_tbp = -1).  mapf_zero_rows_from clears rows [*first_dev, last) of its buffers and nothing in front."""
    from mapf_rl_amd._lib import ERR_INVALID_ARG, check, lib

    B, T, N = 12, 9, 6
    comm, steps, g = _random_windows(B, T, N, 0.3, 5, False)
    rel, slot, order, nact, cnt, nag = _plan_mark(comm, steps)
    obs = torch.rand((B, T, N, 486), device="cuda", generator=g).to(torch.bfloat16)
    hidden = (torch.randn((B * N, 256), device="cuda", generator=g) * 0.3).to(torch.float16)
    Nc, rows = 16, int(cnt.sum())
    pad = rows + 37
    cm = comm.view(torch.uint8)
    dup = torch.zeros((T, B, N), dtype=torch.uint8, device="cuda")
    for t in range(T):
        dup[t] = t  # every entry is the first appearance of its observation: distinct rows == rows
    ucnt = cnt.clone()

    def run(padded):
        n = pad if padded else rows
        out = dict(gidx=torch.empty((T, B, Nc), dtype=torch.int32, device="cuda"), comm_c=torch.empty((T, B, Nc, Nc), dtype=torch.uint8, device="cuda"),
                   h0_c=torch.empty((B, Nc, 256), dtype=torch.bfloat16, device="cuda"), obs_rows=torch.full((n, 486), -7.0, dtype=torch.bfloat16, device="cuda"),
                   row_src=torch.full((n,), -5, dtype=torch.int64, device="cuda"), umap=torch.full((n,), -5, dtype=torch.int32, device="cuda"),
                   row_tbp=torch.full((n,), -5, dtype=torch.int32, device="cuda"))
        args = [T, B, N, Nc, _p(order), _p(nact), _p(cnt), _p(nag), _p(cm), cm.stride(0), cm.stride(1), _p(hidden), 0, _p(obs), obs.stride(0), obs.stride(1),
                _p(out["gidx"]), _p(out["comm_c"]), _p(out["h0_c"]), n, _p(out["row_src"]), _p(out["obs_rows"]), _p(dup), _p(ucnt), _p(out["umap"]), _p(out["row_tbp"])]
        if padded:
            check(lib.mapf_plan_rows_padded(*args, n, n, None), "mapf_plan_rows_padded")
        else:
            check(lib.mapf_plan_rows(*args, None), "mapf_plan_rows")
        return out

    a, b = run(False), run(True)
    for k in ("gidx", "comm_c", "h0_c"):
        assert torch.equal(a[k], b[k]), k
    for k in ("obs_rows", "row_src", "umap", "row_tbp"):
        assert torch.equal(a[k], b[k][:rows]), k
    assert bool((b["row_src"][rows:] == 0).all()) and bool((b["umap"][rows:] == 0).all()) and bool((b["row_tbp"][rows:] == -1).all())
    assert torch.equal(b["obs_rows"][rows:], obs.reshape(-1, 486)[0].expand(pad - rows, 486))
    # mapf_zero_rows_from
    import ctypes

    x, y = torch.ones((50, 64), dtype=torch.bfloat16, device="cuda"), torch.ones((50, 8), dtype=torch.float32, device="cuda")
    ptrs, rb = (ctypes.c_void_p * 2)(x.data_ptr(), y.data_ptr()), (ctypes.c_int * 2)(128, 32)
    for first in (0, 17, 50, 77, -3):
        x.fill_(1), y.fill_(1)
        f = torch.tensor([first], dtype=torch.int32, device="cuda")
        check(lib.mapf_zero_rows_from(ptrs, rb, 2, _p(f), 50, None), "mapf_zero_rows_from")
        lo = min(max(first, 0), 50)
        assert bool((x[:lo] == 1).all()) and bool((x[lo:] == 0).all()) and bool((y[:lo] == 1).all()) and bool((y[lo:] == 0).all()), first
    assert lib.mapf_zero_rows_from(ptrs, rb, 2, None, 50, None) == ERR_INVALID_ARG


def test_rows_scatter_round_trip():
    from mapf_rl_amd._lib import check, lib

    g = torch.Generator(device="cuda").manual_seed(0)
    R, n = 1000, 300
    idx = torch.full((R,), -1, dtype=torch.int32, device="cuda")
    pos = torch.randperm(R, device="cuda", generator=g)[:n]
    idx[pos] = torch.arange(n, dtype=torch.int32, device="cuda")
    rows = torch.randn((n, 768), device="cuda", generator=g).to(torch.bfloat16)
    dense = torch.empty((R, 768), dtype=torch.bfloat16, device="cuda")
    check(lib.mapf_rows_scatter(_p(rows), _p(idx), _p(dense), R, 1536, 1, None), "mapf_rows_scatter")
    want = torch.zeros((R, 768), dtype=torch.bfloat16, device="cuda")
    want[pos] = rows
    assert torch.equal(dense, want)
    back = torch.zeros_like(rows)
    check(lib.mapf_rows_scatter(_p(back), _p(idx), _p(dense), R, 1536, 0, None), "mapf_rows_scatter")
    assert torch.equal(back, rows)


def test_recurrence_pack_kernel_equals_torch_statement():
    from mapf_rl_amd.fused import PackedRecurrence, pack_recurrence_torch, pack_recurrence_transposed, pack_recurrence_transposed_torch, recurrence_params
    from mapf_rl_amd.model import Network

    torch.manual_seed(3)
    net = Network().cuda()
    for p in net.parameters():
        p.data.normal_(0, 0.3)
    w, b = PackedRecurrence().get(net)
    w_ref, b_ref = pack_recurrence_torch(recurrence_params(net))
    assert torch.equal(w, w_ref) and torch.equal(b, b_ref)
    assert torch.equal(pack_recurrence_transposed(recurrence_params(net)), pack_recurrence_transposed_torch(recurrence_params(net)))


def test_encoder_pack_reads_channels_last_weights_in_place():
    from mapf_rl_amd.fused import PackedEncoder, encoder_convs, pack_encoder_backward
    from mapf_rl_amd.model import Network

    torch.manual_seed(4)
    net = Network().cuda()  # convolution weights are stored channels_last
    assert all(c.weight.is_contiguous(memory_format=torch.channels_last) for c in encoder_convs(net.obs_encoder))
    wp, bp = PackedEncoder().get(net.obs_encoder)
    wpt = pack_encoder_backward(net.obs_encoder)
    plain = Network().cuda()
    plain.load_state_dict(net.state_dict())
    plain.obs_encoder.to(memory_format=torch.contiguous_format)
    for c in encoder_convs(plain.obs_encoder):
        c.weight.data = c.weight.data.contiguous()
    assert not encoder_convs(plain.obs_encoder)[1].weight.is_contiguous(memory_format=torch.channels_last)
    wp2, bp2 = PackedEncoder().get(plain.obs_encoder)
    assert torch.equal(wp, wp2) and torch.equal(bp, bp2) and torch.equal(wpt, pack_encoder_backward(plain.obs_encoder))


def _head_reference(a0, a0_tg, a0_on2, bt, steps, action, reward, done, weights, head, head_t):
    """worker.py:296-310,341-344 + model.py:259-262 in fp32 with autograd."""
    B = a0.shape[1]
    ar = torch.arange(B, device=a0.device)
    head = [h.detach().clone().requires_grad_(True) for h in head]

    def q_of(x, h):
        adv = x @ h[0].t() + h[1]
        return x @ h[2].t() + h[3] + adv - adv.mean(1, keepdim=True)

    x = a0.float().detach().requires_grad_(True)
    q_on = q_of(x[bt - 1, ar], head)
    nxt = bt + steps.long() - 1
    q_tg = q_of(a0_tg.float()[nxt, ar], head_t)
    pick = (q_of(a0_on2.float()[nxt, ar], head) if a0_on2 is not None else q_tg).argmax(1, keepdim=True)
    q_next = (1 - done.view(-1, 1)) * q_tg.gather(1, pick)
    q = q_on.gather(1, action.view(-1, 1))
    td = q - (reward.view(-1, 1) + 0.99 ** steps.view(-1, 1) * q_next.detach())
    a = td.abs()
    flag = (a < 1).float()
    loss = (weights.view(-1, 1) * (flag * a.pow(2) * 0.5 + (1 - flag) * (a - 0.5))).mean()
    grads = torch.autograd.grad(loss, [x] + list(head))
    return q.view(-1), q_next.view(-1), td.view(-1).detach(), loss.detach(), grads


@pytest.mark.parametrize("double_q", [False, True])
def test_head_loss_kernel_vs_pytorch(double_q):
    from mapf_rl_amd._lib import check, lib

    g = torch.Generator(device="cuda").manual_seed(5)
    B, To, Tt = 192, 16, 18
    rn = lambda *s: torch.randn(s, device="cuda", generator=g)
    a0, a0_tg = (rn(To, B, 256) * 0.5).to(torch.bfloat16), (rn(Tt, B, 256) * 0.5).to(torch.bfloat16)
    a0_on2 = (rn(Tt, B, 256) * 0.5).to(torch.bfloat16) if double_q else None
    bt = torch.randint(1, To + 1, (B,), device="cuda", generator=g)
    steps = torch.randint(1, 3, (B,), device="cuda", generator=g).float()
    action = torch.randint(0, 5, (B,), device="cuda", generator=g)
    reward = torch.tensor([-0.075, -0.5, 3.0], device="cuda")[torch.randint(0, 3, (B,), device="cuda", generator=g)] * 4  # some |td| > 1
    done = (torch.rand(B, device="cuda", generator=g) < 0.2).float()
    weights = torch.rand(B, device="cuda", generator=g) + 0.1
    head = [rn(5, 256) * 0.2, rn(5) * 0.1, rn(1, 256) * 0.2, rn(1) * 0.1]
    head_t = [rn(5, 256) * 0.2, rn(5) * 0.1, rn(1, 256) * 0.2, rn(1) * 0.1]
    outs = torch.empty((3, B), device="cuda")
    prio = torch.empty(B, dtype=torch.float64, device="cuda")
    loss = torch.empty(1, device="cuda")
    scratch = torch.empty(9 * B, device="cuda")
    d_a0 = torch.empty((To, B, 256), dtype=torch.bfloat16, device="cuda")
    hg = [torch.ones_like(h) for h in head]  # accumulated into
    arr = lambda ts: (ctypes.c_void_p * 4)(*[t.data_ptr() for t in ts])
    check(lib.mapf_dqn_head_loss(B, To, Tt, _p(a0), _p(a0_tg), _p(a0_on2), _p(bt), _p(steps), _p(action), _p(reward), _p(done), _p(weights), arr(head),
                                 arr(head_t), 0.99, _p(outs[0]), _p(outs[1]), _p(outs[2]), _p(prio), _p(loss), _p(scratch), _p(d_a0), arr(hg), None),
          "mapf_dqn_head_loss")
    q, q_next, td, want_loss, grads = _head_reference(a0, a0_tg, a0_on2, bt, steps, action, reward, done, weights, head, head_t)
    close = lambda a, b, tol=1e-4: bool(((a - b).abs() <= tol * torch.clamp(b.abs(), min=1.0)).all())
    assert close(outs[0], q) and close(outs[1], q_next) and close(outs[2], td)
    assert close(prio.float(), td.abs().clamp(min=1e-6)) and close(loss[0], want_loss)
    assert float((td.abs() > 1).float().mean()) > 0.05  # both Huber branches exercised
    assert float((d_a0.float() - grads[0]).abs().max()) <= 1e-2 * float(grads[0].abs().max())  # bf16 output
    ar = torch.arange(B, device="cuda")
    nz = torch.zeros((To, B), dtype=torch.bool, device="cuda")
    nz[bt - 1, ar] = True
    assert not d_a0[~nz].float().abs().any()
    for got, want in zip(hg, grads[1:]):
        assert close(got - 1, want.view_as(got), 1e-4)


def test_adam_kernel_vs_torch_optim():
    from mapf_rl_amd._lib import check, lib

    g = torch.Generator(device="cuda").manual_seed(6)
    n = 100003
    p0 = torch.randn(n, device="cuda", generator=g)
    ref = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([ref], lr=1e-3)
    p, m, v = p0.clone(), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    pb = torch.empty(n, dtype=torch.bfloat16, device="cuda")
    scratch, norm = torch.empty(256, device="cuda"), torch.empty(1, device="cuda")
    for step in range(1, 6):
        grad = torch.randn(n, device="cuda", generator=g) * (3.0 if step % 2 else 0.01)  # norm ~950 (clipped at 40) / ~3 (not clipped)
        ref.grad = grad.clone()
        want_norm = torch.nn.utils.clip_grad_norm_([ref], 40.0)
        opt.step()
        gg = grad.clone()
        check(lib.mapf_adam_step(n, _p(p), _p(gg), _p(m), _p(v), _p(pb), _p(scratch), _p(norm), 1e-3, 0.9, 0.999, 1e-8, step, 40.0, None), "mapf_adam_step")
        assert abs(float(norm) - float(want_norm)) <= 1e-5 * float(want_norm)
        assert torch.allclose(gg, ref.grad, rtol=1e-5, atol=1e-8)
        assert torch.allclose(p, ref.detach(), rtol=1e-5, atol=1e-6), float((p - ref.detach()).abs().max())
        assert torch.equal(pb, p.to(torch.bfloat16))


def _grads_of(lr):
    out = {}

    def grab(learner):
        for k, p in learner.model.named_parameters():
            out[k] = p.grad.detach().float().clone()

    lr.grad_hook = grab
    return out


@pytest.mark.parametrize("tag", ["b40", "b6", "b128", "upd"])
def test_fused_update_equals_autograd_update(tag):
    """Learner.update through update.FusedUpdate against the same update through autograd over Network.bootstrap (both bf16 kernels
    underneath; the fused path computes the dueling head and the loss in fp32 instead of bf16 autocast): TD errors, loss, gradient
    norm, every parameter's gradient and the parameters after the Adam step."""
    from mapf_rl_amd.learner import Learner

    if tag == "upd":
        z = H.load_npz("dqn_update.npz")
        mk = lambda: _batch(z, "cuda", torch.bfloat16)
    else:
        z = H.load_npz("dqn_big.npz")
        mk = lambda: BG.batch(z, tag, "cuda", torch.bfloat16)
    res = {}
    for fused in (False, True):
        Learner.FUSED_UPDATE = fused
        try:
            lr = _models("cuda")
        finally:
            Learner.FUSED_UPDATE = True
        assert (lr._fused is not None) == fused
        grads = _grads_of(lr)
        out = lr.update(mk())
        torch.cuda.synchronize()
        res[fused] = (out, grads, {k: v.detach().clone() for k, v in lr.model.state_dict().items()})
    (o0, g0, s0), (o1, g1, s1) = res[False], res[True]
    for k in ("td", "q", "q_next"):
        a, b = o0[k].float().view(-1), o1[k].float().view(-1)
        assert bool(((a - b).abs() <= 2e-2 * torch.clamp(a.abs(), min=1.0)).all()), (k, float((a - b).abs().max()))
    assert abs(float(o0["loss"]) - float(o1["loss"])) <= 2e-2 * max(1.0, float(o0["loss"]))
    assert abs(float(o0["grad_norm"]) - float(o1["grad_norm"])) <= 3e-2 * float(o0["grad_norm"])
    tot = float(torch.sqrt(sum((v ** 2).sum() for v in g0.values())))
    assert set(g0) == set(g1) and len(g0) == 35
    for k in g0:
        d, n = float((g0[k] - g1[k]).norm()), float(g0[k].norm())
        assert d <= 5e-2 * n + 2e-3 * tot, (k, d, n)
    for k in s0:  # one Adam step of size 1e-4 from identical weights
        assert torch.allclose(s0[k], s1[k], rtol=0, atol=2.5e-4), k


@pytest.mark.parametrize("tag", ["b6", "upd"])
def test_windows_sharing_a_recurrence_tile_give_the_same_update(tag):
    """FusedUpdate.TILE_WINDOWS: two (four) windows of <= 8 (<= 4) agents that matter share a 16-row tile of the recurrence kernels under a
    block-diagonal mask (mapf_plan_rows at compact width 8 / 4, mapf_recurrent_*_packed).  Row by row the kernels compute the same bits,
    so Q-values, TD errors, loss and every weight gradient are IDENTICAL to one window per tile; the recurrence's bias gradients are sums
    over tiles instead of windows (same terms, another order)."""
    from mapf_rl_amd.update import FusedUpdate

    if tag == "upd":
        z = H.load_npz("dqn_update.npz")
        mk = lambda: _batch(z, "cuda", torch.bfloat16)
    else:
        z = H.load_npz("dqn_big.npz")
        mk = lambda: BG.batch(z, tag, "cuda", torch.bfloat16)
    res, widths = {}, {}
    try:
        for tiles in (False, True):
            FusedUpdate.TILE_WINDOWS = tiles
            lr = _models("cuda")
            grads = _grads_of(lr)
            batch = mk()
            pl = lr._fused._finish_plan(lr._fused.plan(batch))
            widths[tiles] = (pl["online"].nc, pl["target"].nc, pl["online"].B)
            out = lr.update(batch)
            torch.cuda.synchronize()
            res[tiles] = (out, grads)
    finally:
        FusedUpdate.TILE_WINDOWS = True
    assert widths[False][0] == 16 and widths[False][1] == 16
    if widths[True][2] % 2 == 0:
        assert widths[True][0] < 16 and widths[True][1] < 16, widths  # (the case must exercise what it names)
    (o0, g0), (o1, g1) = res[False], res[True]
    for k in ("td", "q", "q_next"):
        assert torch.equal(o0[k], o1[k]), k
    assert float(o0["loss"]) == float(o1["loss"])
    for k in g0:
        if "bias" in k and ("recurrent" in k or "comm" in k):
            assert torch.allclose(g0[k], g1[k], rtol=2e-3, atol=1e-5 + 1e-3 * float(g0[k].abs().max())), k
        else:
            assert torch.equal(g0[k], g1[k]), k


def test_fused_update_pruning_is_dead_code_elimination():
    """The same fused update with every observation of the window encoded (Network.PRUNE_UNREACHABLE = False: mapf_plan_mark's
    mark_all) and with only the entries that can reach agent 0's Q-value: same TD errors and gradients up to the order of sums."""
    from mapf_rl_amd.model import Network

    z = H.load_npz("dqn_big.npz")
    res = {}
    try:
        for prune in (True, False):
            Network.PRUNE_UNREACHABLE = prune
            lr = _models("cuda")
            grads = _grads_of(lr)
            out = lr.update(BG.batch(z, "b40", "cuda", torch.bfloat16))
            torch.cuda.synchronize()
            res[prune] = (out, grads)
    finally:
        Network.PRUNE_UNREACHABLE = True
    (o0, g0), (o1, g1) = res[False], res[True]
    a, b = o0["td"].view(-1), o1["td"].view(-1)
    assert bool(((a - b).abs() <= 1e-2 * torch.clamp(a.abs(), min=1.0)).all()), float((a - b).abs().max())
    tot = float(torch.sqrt(sum((v ** 2).sum() for v in g0.values())))
    for k in g0:
        assert float((g0[k] - g1[k]).norm()) <= 2e-2 * float(g0[k].norm()) + 1e-3 * tot, k


def test_fused_update_survives_state_dict_round_trip(tmp_path):
    """Parameters are views of flat buffers: state_dict / load_state_dict / save keep the reference's keys and the bf16 copy follows."""
    from mapf_rl_amd.learner import Learner
    from mapf_rl_amd.model import Network

    z = H.load_npz("dqn_update.npz")
    lr = _models("cuda")
    sd0 = {k: v.detach().clone() for k, v in lr.model.state_dict().items()}
    out0 = lr.update(_batch(z, "cuda", torch.bfloat16))
    path = lr.save(str(tmp_path / "m.pth"))
    assert set(torch.load(path).keys()) == set(Network().state_dict().keys())
    lr.load_state_dict(sd0)  # back to the initial weights (target network too): the same update must come out again
    lr.tar_model.load_state_dict(_models("cpu").tar_model.state_dict())
    lr._fused.flat.exp_avg.zero_(), lr._fused.flat.exp_avg_sq.zero_()
    lr._fused.flat.step = 0
    out1 = lr.update(_batch(z, "cuda", torch.bfloat16))
    assert torch.equal(out0["td"], out1["td"]) and float(out0["loss"]) == float(out1["loss"])


@pytest.mark.parametrize("B,T,N,p", [(24, 16, 40, 0.05), (6, 18, 16, 0.3), (5, 4, 3, 0.5)])
def test_recurrence_kernels_compact_rows_equal_dense_rows(B, T, N, p):
    """mapf_recurrent_forward_save / _infer / _backward with compact rows (row_index = mapf_plan_rows' gidx: only the entries that
    can reach agent 0's Q-value have a row in gi, the saved tensors and the gradient outputs) against the same launches on dense
    [T][E][N] rows: agent-0 states and every row that exists are the same bits."""
    from mapf_rl_amd._lib import check, lib
    from mapf_rl_amd.fused import PackedRecurrence, pack_recurrence_transposed, recurrence_params
    from mapf_rl_amd.model import Network

    torch.manual_seed(B)
    net = Network().cuda()
    w, b = PackedRecurrence().get(net)
    wt = pack_recurrence_transposed(recurrence_params(net))
    comm, steps, g = _random_windows(B, T, N, p, 5 + N)
    rel, slot, order, nact, cnt, nag = _plan_mark(comm, steps)
    Nc = 16 * -(-int(nag.max()) // 16)
    rows = int(cnt.sum())
    gidx = torch.empty((T, B, Nc), dtype=torch.int32, device="cuda")
    comm_c = torch.empty((T, B, Nc, Nc), dtype=torch.uint8, device="cuda")
    h0_c = torch.empty((B, Nc, 256), dtype=torch.bfloat16, device="cuda")
    hidden = (torch.randn((B * N, 256), device="cuda", generator=g) * 0.3).to(torch.float16)
    cm = comm.view(torch.uint8)
    check(lib.mapf_plan_rows(T, B, N, Nc, _p(order), _p(nact), _p(cnt), _p(nag), _p(cm), cm.stride(0), cm.stride(1), _p(hidden), 0, None, 0, 0,
                             _p(gidx), _p(comm_c), _p(h0_c), 0, None, None, None, None, None, None, None), "mapf_plan_rows")
    gi_rows = (torch.randn((rows, 768), device="cuda", generator=g) * 0.5).to(torch.bfloat16)
    gi_dense = torch.empty((T, B, Nc, 768), dtype=torch.bfloat16, device="cuda")
    check(lib.mapf_rows_scatter(_p(gi_rows), _p(gidx), _p(gi_dense), T * B * Nc, 1536, 1, None), "mapf_rows_scatter")
    d_a0 = (torch.randn((T, B, 256), device="cuda", generator=g) * 0.1).to(torch.bfloat16)
    # (an entry without a row must not receive an external gradient: behind the window's last step agent 0 has none)
    d_a0 = d_a0 * (torch.arange(T, device="cuda").view(T, 1, 1) < steps.view(1, B, 1)).to(torch.bfloat16)
    bf = torch.bfloat16
    widths_s = [(1, 256), (1, 1024), (2, 256), (2, 384), (2, 128), (2, 64), (2, 1024)]
    widths_o = [(1, 768), (1, 768), (2, 768), (2, 768), (2, 64), (2, 384)]

    def run(compact):
        R = rows if compact else T * B * Nc
        saves = [torch.zeros((k, R, wd), dtype=bf, device="cuda") for k, wd in widths_s] + [torch.zeros((2, T * B, 2, 48, 64), dtype=bf, device="cuda")]
        outs = [torch.zeros((k, R, wd), dtype=bf, device="cuda") for k, wd in widths_o] + [torch.zeros((B, 2432), dtype=torch.float32, device="cuda")]
        h_out = torch.empty((B, Nc, 256), dtype=bf, device="cuda")
        a0 = torch.empty((T, B, 256), dtype=bf, device="cuda")
        a0i = torch.empty((T, B, 256), dtype=bf, device="cuda")
        sp = (ctypes.c_void_p * 8)(*[t.data_ptr() for t in saves])
        op = (ctypes.c_void_p * 7)(*[t.data_ptr() for t in outs])
        gi = gi_rows if compact else gi_dense
        ri, nr = (_p(gidx), rows) if compact else (None, 0)
        check(lib.mapf_recurrent_infer(_p(gi), _p(h0_c), _p(comm_c), _p(w), _p(b), T, B, Nc, _p(h_out), _p(a0i), ri, nr, None), "mapf_recurrent_infer")
        check(lib.mapf_recurrent_forward_save(_p(gi), _p(h0_c), _p(comm_c), _p(w), _p(b), T, B, Nc, _p(h_out), _p(a0), sp, ri, nr, None),
              "mapf_recurrent_forward_save")
        check(lib.mapf_recurrent_backward(sp, _p(comm_c), _p(d_a0), _p(wt), T, B, Nc, op, ri, nr, None), "mapf_recurrent_backward")
        torch.cuda.synchronize()
        return a0i, a0, saves, outs

    a0i_d, a0_d, sv_d, out_d = run(False)
    a0i_c, a0_c, sv_c, out_c = run(True)
    assert torch.equal(a0_d, a0_c) and torch.equal(a0i_d, a0i_c) and torch.equal(a0_c, a0i_c)
    sel = (gidx.view(-1) >= 0).nonzero().view(-1)          # dense rows that exist, in row order? no: gather through gidx
    dense_of_row = torch.empty(rows, dtype=torch.int64, device="cuda")
    dense_of_row[gidx.view(-1)[sel].long()] = sel
    for k, (d, c) in enumerate(zip(sv_d[:7] + out_d[:6], sv_c[:7] + out_c[:6])):
        bad = (d[:, dense_of_row] != c).any(dim=2).nonzero()
        assert bad.numel() == 0, (k, bad[:8].tolist(), gidx.view(-1, Nc)[:3].tolist(), nact[:, 0].tolist())
    assert torch.equal(sv_d[7], sv_c[7])
    assert torch.allclose(out_d[6], out_c[6], rtol=1e-5, atol=1e-6)  # bias column sums (same order of addition; zero rows skipped)
    # rows that do not exist carry no gradient in the dense launch either: pruning is dead-code elimination
    mask = torch.ones(T * B * Nc, dtype=torch.bool, device="cuda")
    mask[sel] = False
    for d in out_d[:6]:
        assert not d[:, mask].float().abs().any()


@pytest.mark.parametrize("B,T,N,p", [(24, 18, 40, 0.05), (4, 18, 128, 0.02), (7, 5, 3, 0.5)])
def test_repeated_observations_are_found_numbered_and_summed(B, T, N, p):
    """mapf_obs_dup / mapf_plan_rows(dup) / mapf_dedup_sum: an entry that carries the same 486 values as the same agent at ANY earlier
    step of the window (consecutive or not: an agent that comes back to a cell whose neighbourhood looks the same) shares that step's
    row of distinct observations -- obs_rows[umap[r]] is entry r's observation for EVERY entry --, the distinct rows are counted per
    window for both closures, and the gradient of a shared row is the sum over its entries."""
    from mapf_rl_amd._lib import check, lib

    comm, steps, g = _random_windows(B, T, N, p, 21 + N, True)
    To = T - 2
    bt = torch.clamp(steps, max=To)
    extra = torch.randint(1, 3, (B,), device="cuda", generator=g).float()
    cm = comm.view(torch.uint8)
    plans = []
    ucnt = torch.full((2, B), 77, dtype=torch.int32, device="cuda")  # (zeroed by plan_mark)
    for k, (Tk, ex) in enumerate(((To, None), (T, extra))):
        slot = torch.empty((B, N), dtype=torch.int16, device="cuda")
        order = torch.empty((B, N), dtype=torch.int16, device="cuda")
        nact = torch.empty((Tk, B), dtype=torch.int32, device="cuda")
        cnt = torch.empty(B, dtype=torch.int32, device="cuda")
        nag = torch.empty(B, dtype=torch.int32, device="cuda")
        rel = torch.empty((Tk, B, N), dtype=torch.uint8, device="cuda")
        check(lib.mapf_plan_mark(_p(cm), cm.stride(0), cm.stride(1), _p(bt), _p(ex), Tk, B, N, 0, _p(rel), _p(slot), _p(order), _p(nact), _p(cnt), _p(nag),
                                 _p(ucnt[k]), None), "mapf_plan_mark")
        plans.append((Tk, slot, order, nact, cnt, nag, rel))
    obs = torch.rand((T, B, N, 486), device="cuda", generator=g).to(torch.bfloat16)
    rep = torch.rand((T, B, N), device="cuda", generator=g) < 0.5
    back = torch.randint(1, 6, (T, B, N), device="cuda", generator=g)
    for t in range(1, T):  # the observation of the same agent 1..5 steps earlier again (runs, and repeats with other observations in between)
        src = torch.clamp(t - back[t], min=0)
        old = torch.gather(obs[:t].permute(1, 2, 0, 3), 2, src.view(B, N, 1, 1).expand(B, N, 1, 486)).squeeze(2)
        obs[t] = torch.where(rep[t].unsqueeze(-1), old, obs[t])
    obs = obs.transpose(0, 1)  # [B, T, N, 486] view of time-major memory
    dup = torch.empty((T, B, N), dtype=torch.uint8, device="cuda")
    check(lib.mapf_obs_dup(T, To, B, N, _p(obs), obs.stride(0), obs.stride(1), _p(plans[0][1]), _p(plans[1][1]), _p(plans[0][3]), _p(plans[1][3]), _p(dup),
                           _p(ucnt[0]), _p(ucnt[1]), None), "mapf_obs_dup")
    torch.cuda.synchronize()
    ot = obs.transpose(0, 1)                                          # [T, B, N, 486]
    eq = (ot.unsqueeze(1) == ot.unsqueeze(0)).all(dim=-1)             # eq[t, t0]: same observation at steps t and t0
    tri = torch.tril(torch.ones((T, T), dtype=torch.bool, device="cuda"))
    first = (eq & tri.view(T, T, 1, 1)).float().argmax(dim=1)          # the first step t0 <= t with the same values
    rel_t = plans[1][6].bool()
    ar = torch.arange(T, device="cuda").view(T, 1, 1)
    assert torch.equal(dup.long()[rel_t], first[rel_t]) and torch.equal(dup.long()[~rel_t], ar.expand(T, B, N)[~rel_t])
    assert int(((first < ar - 1) & rel_t).sum()) > 0                  # (non-consecutive repeats are present)
    for k in range(2):
        Tk, rel = plans[k][0], plans[k][6].bool()
        assert torch.equal(ucnt[k].long(), (rel & (first[:Tk] == ar[:Tk])).sum(dim=(0, 2)))
    hidden = torch.zeros((B * N, 256), dtype=torch.float16, device="cuda")
    for k in range(2):
        Tk, slot, order, nact, cnt, nag, rel = plans[k]
        Nc = 16 * -(-int(nag.max()) // 16)
        rows, urows = int(cnt.sum()), int(ucnt[k].sum())
        gidx = torch.empty((Tk, B, Nc), dtype=torch.int32, device="cuda")
        comm_c = torch.empty((Tk, B, Nc, Nc), dtype=torch.uint8, device="cuda")
        h0_c = torch.empty((B, Nc, 256), dtype=torch.bfloat16, device="cuda")
        obs_u = torch.empty((urows, 486), dtype=torch.bfloat16, device="cuda")
        row_src = torch.empty(urows, dtype=torch.int64, device="cuda")
        umap = torch.empty(rows, dtype=torch.int32, device="cuda")
        tbp = torch.empty(rows, dtype=torch.int32, device="cuda")
        check(lib.mapf_plan_rows(Tk, B, N, Nc, _p(order), _p(nact), _p(cnt), _p(nag), _p(cm), cm.stride(0), cm.stride(1), _p(hidden), 0, _p(obs),
                                 obs.stride(0), obs.stride(1), _p(gidx), _p(comm_c), _p(h0_c), urows, _p(row_src), _p(obs_u), _p(dup), _p(ucnt[k]), _p(umap),
                                 _p(tbp), None), "mapf_plan_rows")
        torch.cuda.synchronize()
        # every entry finds its own observation in the distinct rows, and the distinct rows are all used, in order of first use
        G, O = gidx.cpu().numpy(), order.cpu().numpy().astype(np.int64)
        um, tb = umap.cpu().numpy(), tbp.cpu().numpy()
        ob, ou = obs.float().cpu().numpy(), obs_u.float().cpu().numpy()
        assert um.min() == 0 and um.max() == urows - 1 and len(np.unique(um)) == urows
        first = np.sort(np.unique(um, return_index=True)[1])
        assert np.array_equal(um[first], np.arange(urows))
        for r in range(0, rows, max(1, rows // 400)):
            t, i, b = (tb[r] >> 24) & 31, (tb[r] >> 16) & 255, tb[r] & 0xFFFF   # (bit 30: the entry reuses an earlier step's row)
            assert G[t, b, i] == r and np.array_equal(ou[um[r]], ob[b, t, O[b, i]])
        # gradient of a shared row
        d_rows = (torch.randn((rows, 768), device="cuda", generator=g) * 0.3).to(torch.bfloat16)
        d_u = torch.empty((urows, 768), dtype=torch.bfloat16, device="cuda")
        check(lib.mapf_dedup_sum(Tk, B, Nc, rows, 1536, _p(gidx), _p(umap), _p(tbp), _p(d_rows), _p(d_u), None), "mapf_dedup_sum")
        want = torch.zeros((urows, 768), device="cuda").index_add_(0, umap.long(), d_rows.float())
        assert torch.allclose(d_u.float(), want, rtol=1e-2, atol=1e-2) and float((d_u.float() - want).abs().max()) <= 0.02 * float(want.abs().max())


def test_fused_update_with_and_without_observation_reuse():
    """update.FusedUpdate.DEDUP: encoding every entry or the distinct observations only gives the same forward bits (the latent of a
    repeated observation is a copy) and the same gradients up to the order of sums."""
    from mapf_rl_amd.update import FusedUpdate

    z = H.load_npz("dqn_big.npz")
    res = {}
    try:
        for dedup in (False, True):
            FusedUpdate.DEDUP = dedup
            lr = _models("cuda")
            grads = _grads_of(lr)
            b = BG.batch(z, "b40", "cuda", torch.bfloat16)
            obs = b[0].clone()
            obs[:, 5:9] = obs[:, 4:5]   # steps 4..8 of every window: nobody's observation changes -> runs of 5
            out = lr.update((obs,) + b[1:])
            torch.cuda.synchronize()
            res[dedup] = (out, grads, lr._fused)
    finally:
        FusedUpdate.DEDUP = True
    (o0, g0, _), (o1, g1, _) = res[False], res[True]
    assert torch.equal(o0["td"], o1["td"]) and torch.equal(o0["q"], o1["q"]) and float(o0["loss"]) == float(o1["loss"])
    tot = float(torch.sqrt(sum((v ** 2).sum() for v in g0.values())))
    for k in g0:
        assert float((g0[k] - g1[k]).norm()) <= 1e-2 * float(g0[k].norm()) + 1e-3 * tot, k
