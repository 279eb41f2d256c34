"""GPU (bf16 autocast): the product Network against reference goldens.  Tolerance |dQ| <= 2e-2*max(1,|Q|)
(SURVEY.md 8(c)); greedy-action agreement is asserted only on rows whose top-2 gap exceeds the tolerance."""
import numpy as np
import pytest
import torch

from tests import helpers as H
from tests.test_model_cpu import _net

pytestmark = pytest.mark.gpu
TOL = 2e-2


def _close(a, b, tol=TOL):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.all(np.abs(a - b) <= tol * np.maximum(1.0, np.abs(b)))


@pytest.mark.parametrize("nag", [16, 32])
def test_step_bf16(nag):
    z = H.load_npz("dqn_model.npz")
    net, _ = _net()
    net.cuda()
    pre = "step%d_" % nag
    net.reset()
    for t in range(z[pre + "q"].shape[0]):
        obs = H.unpack_bits(z[pre + "obs_bits"][t], (nag, 6, 9, 9))
        actions, q, hidden, cm = net.step(torch.from_numpy(obs), torch.from_numpy(z[pre + "pos"][t].astype(np.int64)),
                                          comm_mask=z[pre + "comm_mask"][t])
        assert _close(q, z[pre + "q"][t]), (t, np.abs(q - z[pre + "q"][t]).max())
        assert _close(hidden, z[pre + "hidden"][t], 3e-2)
        srt = np.sort(z[pre + "q"][t], axis=1)
        clear = (srt[:, -1] - srt[:, -2]) > 2 * TOL * np.maximum(1, np.abs(srt[:, -1]))
        assert np.array_equal(np.array(actions)[clear], z[pre + "actions"][t][clear])


def test_bootstrap_bf16_and_fp32_on_gpu():
    z = H.load_npz("dqn_model.npz")
    net, _ = _net()
    net.cuda()
    B, T, A = z["boot_shape"]
    obs = torch.from_numpy(H.unpack_bits(z["boot_obs_bits"], (B, T, A, 6, 9, 9))).cuda()
    args = (torch.from_numpy(z["boot_steps"]).cuda(), torch.from_numpy(z["boot_hidden"]).cuda(), torch.from_numpy(z["boot_comm"]).cuda())
    with torch.no_grad():
        q = net.bootstrap(obs.to(torch.bfloat16), *args).cpu().numpy()
    assert _close(q, z["boot_q"]), np.abs(q - z["boot_q"]).max()


def test_batched_step_vs_env_observations():
    """E envs x N agents straight from the HIP environment (uint8 obs, int16 pos) through step_batch."""
    import mapf_rl_amd as M
    from mapf_rl_amd.model import comm_mask_from_pos

    net, _ = _net()
    net.cuda()
    E, L, N = 16, 32, 40
    maps, agents, goals, _ = M.generate_scenarios(E, L, N, 0.3, seed=4)
    env = M.VecEnvironment(E, L, N)
    env.load(maps, agents, goals)
    obs, pos = env.observe()
    a, q, h, cm = net.step_batch(obs, pos, None)
    assert a.shape == (E, N) and q.shape == (E, N, 5) and h.shape == (E * N, 256) and cm.shape == (E, N, N)
    assert torch.equal(cm, comm_mask_from_pos(pos))
    # same inputs through the CPU fp32 path
    net_c, _ = _net()
    a2, q2, h2, cm2 = net_c.step_batch(obs.cpu(), pos.cpu(), None)
    assert torch.equal(cm.cpu(), cm2)
    assert _close(q.cpu().numpy(), q2.numpy())
    obs2, pos2, *_ = env.step(a.to(torch.int8))
    env.check_status()
    a3, q3, h3, _ = net.step_batch(obs2, pos2, h)
    assert torch.isfinite(q3).all()


def test_fast_recurrence_matches_module_path():
    """`Network._recur_fast` (hoisted input projection, fused QKV, weight gradients deferred to one GEMM per weight
    at the end of backward) against the plain module loop of `bootstrap` at the same bf16 precision: Q-values and
    every parameter gradient."""
    from mapf_rl_amd.model import Network

    torch.manual_seed(3)
    net = Network().cuda()
    with torch.no_grad():  # non-zero biases so that their deferred gradients are exercised
        for p in net.parameters():
            if p.dim() == 1:
                p.uniform_(-0.1, 0.1)
    B, T, N = 6, 5, 7
    g = torch.Generator(device="cuda").manual_seed(4)
    obs = (torch.rand((B, T, N, 6, 9, 9), device="cuda", generator=g) < 0.3).to(torch.bfloat16)
    steps = torch.randint(1, T + 1, (B,), device="cuda", generator=g)
    hidden = (torch.randn((B * N, 256), device="cuda", generator=g) * 0.3).to(torch.bfloat16)
    comm = torch.rand((B, T, N, N), device="cuda", generator=g) < 0.3
    comm |= torch.eye(N, dtype=torch.bool, device="cuda")
    comm[0, :, 1] = torch.eye(N, dtype=torch.bool, device="cuda")[1]   # an agent without partners (no update)
    r = torch.randn((B, 5), device="cuda", generator=g)
    res = {}
    Network.FUSED_BPTT = False  # this test is about the PyTorch-level fast path; the kernels have their own below
    for fast in (True, False):
        Network.FAST_RECURRENCE = fast
        try:
            net.zero_grad()
            q = net.bootstrap(obs, steps, hidden, comm)
            (q * r).sum().backward()
            res[fast] = (q.detach().clone(), {k: p.grad.detach().clone() for k, p in net.named_parameters()})
            with torch.no_grad():
                q_ng = net.bootstrap(obs, steps, hidden, comm)
            assert torch.allclose(q_ng, q.detach(), rtol=3e-2, atol=3e-2)
        finally:
            Network.FAST_RECURRENCE = True
    Network.FUSED_BPTT = True
    assert torch.allclose(res[True][0], res[False][0], rtol=3e-2, atol=3e-2)
    scale = max(float(b.norm()) for b in res[False][1].values())
    for k in res[True][1]:
        a, b = res[True][1][k], res[False][1][k]
        # W_K.bias has an exactly-zero true gradient (softmax is shift-invariant): both sides are rounding noise,
        # hence the absolute floor relative to the largest parameter gradient
        assert float((a - b).norm()) <= 6e-2 * float(b.norm()) + 1e-4 * scale, (k, float((a - b).norm()), float(b.norm()))


@pytest.mark.parametrize("E,N,T", [(5, 7, 1), (3, 40, 4), (2, 48, 2), (4, 1, 3), (3, 17, 2), (3, 16, 3), (2, 15, 2), (2, 32, 2), (3, 24, 3), (2, 33, 2),  # <= 16 / <= 32: the one- / two-tile builds
                                   (2, 64, 2), (3, 49, 3), (2, 100, 2), (3, 128, 3), (1, 65, 1)])  # > 48: csrc/mapf_recur_wide.hip
def test_fused_recurrence_kernel_matches_module_path(E, N, T):
    """mapf_recurrent_infer (GRU cell + two communication rounds per step, all T steps in one launch) against the
    PyTorch module path at the same bf16 precision: actor step (T = 1, with and without an incoming hidden state) and
    the no-gradient bootstrap (agent-0 states of every step -> Q-values)."""
    from mapf_rl_amd.model import Network

    torch.manual_seed(E * 100 + N)
    net = Network().cuda()
    with torch.no_grad():
        for p in net.parameters():
            if p.dim() == 1:
                p.uniform_(-0.1, 0.1)
    g = torch.Generator(device="cuda").manual_seed(N)
    obs = (torch.rand((E, T, N, 6, 9, 9), device="cuda", generator=g) < 0.3).to(torch.uint8)
    comm = torch.rand((E, T, N, N), device="cuda", generator=g) < 0.25
    comm |= torch.eye(N, dtype=torch.bool, device="cuda")
    if N > 2:
        comm[0, :, 1] = torch.eye(N, dtype=torch.bool, device="cuda")[1]   # an agent without partners keeps its state
    comm[-1, -1] = False   # a zero-padded window row (worker.py:139-142): every mask row all-False, nobody is updated
    hidden = (torch.randn((E * N, 256), device="cuda", generator=g) * 0.3).to(torch.bfloat16)
    steps = torch.randint(1, T + 1, (E,), device="cuda", generator=g)
    pos = torch.zeros((E, N, 2), dtype=torch.int16, device="cuda")
    out = {}
    for fused in (True, False):
        Network.FUSED_RECURRENCE = fused
        try:
            a = net.step_batch(obs[:, 0], pos, None, comm[:, 0])
            b = net.step_batch(obs[:, 0], pos, hidden, comm[:, 0])
            with torch.no_grad():
                q = net.bootstrap(obs, steps, hidden, comm)
            out[fused] = (a[1], a[2].float(), b[1], b[2].float(), q)
        finally:
            Network.FUSED_RECURRENCE = True
    for x, y in zip(out[True], out[False]):
        assert x.shape == y.shape
        assert torch.allclose(x, y, rtol=3e-2, atol=3e-2), float((x - y).abs().max())


@pytest.mark.parametrize("E,N,T", [(600, 6, 1), (601, 16, 1), (515, 1, 2), (700, 24, 1), (529, 32, 3), (513, 17, 1), (2049, 5, 1)])
def test_two_environments_per_workgroup_step_like_one(E, N, T):
    """With more environments than CUs, the <= 32-agent builds of mapf_recurrent_infer step TWO environments per workgroup (the weight
    fragments a wave fetches serve both: csrc/mapf_recur.hip, recurrent_infer_kernel<false, true>), q | k, ctx and info living in LDS
    bytes that are dead at the time.  Same bits as the one-environment launches the same call makes for <= 256 environments: hidden
    states and agent-0 states, with and without an initial state, odd environment counts (a pair without a second member) included."""
    from mapf_rl_amd import fused

    g = torch.Generator(device="cuda").manual_seed(E + N)
    w = (torch.randn(fused.RECUR_WEIGHT_ELEMS, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
    b = torch.randn(fused.RECUR_BIAS_ELEMS, device="cuda", generator=g) * 0.1
    gi = (torch.randn((T, E, N, 768), device="cuda", generator=g) * 0.5).to(torch.bfloat16)
    h0 = (torch.randn((E, N, 256), device="cuda", generator=g) * 0.3).to(torch.bfloat16)
    comm = torch.rand((T, E, N, N), device="cuda", generator=g) < 0.3
    comm |= torch.eye(N, dtype=torch.bool, device="cuda")
    comm[:, 3] = False  # (a zero-padded row: nobody is updated)
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    assert E > cus, "the launch under test needs more environments than CUs"
    for start in (h0, None):
        whole, a_whole = fused.recurrent_infer(gi, start, comm, w, b, want_agent0=True)
        parts, a_parts = [], []
        for lo in range(0, E, 200):  # <= CUs: one environment per workgroup
            hi = min(E, lo + 200)
            h, a = fused.recurrent_infer(gi[:, lo:hi], None if start is None else start[lo:hi], comm[:, lo:hi], w, b, want_agent0=True)
            parts.append(h)
            a_parts.append(a)
        assert not whole.isnan().any()
        assert torch.equal(whole, torch.cat(parts, 0))
        assert torch.equal(a_whole, torch.cat(a_parts, 1))


@pytest.mark.parametrize("E,N", [(4096, 40), (4096, 6), (4095, 24)])
def test_recurrence_at_full_size_treats_environments_independently(E, N):
    """BASELINE config 2's actor batch (4096 environments x 40 agents) and two smaller-agent batches of the same size: an environment's
    new hidden states do not depend on where it stands in the batch (which workgroup walks it, which environment shares the workgroup,
    the order in which that workgroup's waves fetch the weight tiles) -- the batch in reversed order returns the reversed result, bit for
    bit -- and an environment stepped alone returns the same bits."""
    from mapf_rl_amd import fused

    g = torch.Generator(device="cuda").manual_seed(N)
    w = (torch.randn(fused.RECUR_WEIGHT_ELEMS, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
    b = torch.randn(fused.RECUR_BIAS_ELEMS, device="cuda", generator=g) * 0.1
    gi = (torch.randn((1, E, N, 768), device="cuda", generator=g) * 0.5).to(torch.bfloat16)
    h0 = (torch.randn((E, N, 256), device="cuda", generator=g) * 0.3).to(torch.bfloat16)
    comm = torch.rand((1, E, N, N), device="cuda", generator=g) < 0.15
    comm |= torch.eye(N, dtype=torch.bool, device="cuda")
    out, a0 = fused.recurrent_infer(gi, h0, comm, w, b, want_agent0=True)
    rev, a0r = fused.recurrent_infer(gi.flip(1).contiguous(), h0.flip(0).contiguous(), comm.flip(1).contiguous(), w, b, want_agent0=True)
    assert not out.isnan().any()
    assert torch.equal(out, rev.flip(0)) and torch.equal(a0, a0r.flip(1))
    assert torch.equal(a0[0], out[:, 0])  # agent 0's state after the step is row 0 of its environment
    for e in (0, 1, E // 2, E - 1):
        one, _ = fused.recurrent_infer(gi[:, e:e + 1].contiguous(), h0[e:e + 1].contiguous(), comm[:, e:e + 1].contiguous(), w, b)
        assert torch.equal(one[0], out[e]), e


@pytest.mark.parametrize("E,N", [(320, 5), (600, 20)])
def test_policy_step_of_many_small_environments_equals_its_halves(E, N):
    """Network.step_batch over more environments than CUs (where the recurrence steps two environments per workgroup) returns, for every
    environment, the bits it returns when the same environments are stepped in groups of <= 160 (one environment per workgroup):
    actions, Q-values, hidden states -- and stays within the module path's tolerance."""
    from mapf_rl_amd.model import Network

    torch.manual_seed(E + N)
    net = Network().cuda()
    g = torch.Generator(device="cuda").manual_seed(E)
    obs = (torch.rand((E, N, 6, 9, 9), device="cuda", generator=g) < 0.3).to(torch.uint8)
    comm = torch.rand((E, N, N), device="cuda", generator=g) < 0.3
    comm |= torch.eye(N, dtype=torch.bool, device="cuda")
    hidden = (torch.randn((E * N, 256), device="cuda", generator=g) * 0.3).to(torch.bfloat16)
    pos = torch.zeros((E, N, 2), dtype=torch.int16, device="cuda")
    assert E > torch.cuda.get_device_properties(0).multi_processor_count
    whole = net.step_batch(obs, pos, hidden, comm)
    whole = [x.clone() for x in whole[:3]]
    parts = []
    for lo in range(0, E, 160):
        hi = min(E, lo + 160)
        r = net.step_batch(obs[lo:hi].contiguous(), pos[lo:hi].contiguous(), hidden[lo * N:hi * N].contiguous(), comm[lo:hi].contiguous())
        parts.append([x.clone() for x in r[:3]])
    for k in range(3):
        assert torch.equal(whole[k].reshape(E, -1), torch.cat([p[k].reshape(-1, whole[k].numel() // E) for p in parts], 0)), k
    Network.FUSED_RECURRENCE = False
    try:
        ref = net.step_batch(obs, pos, hidden, comm)
    finally:
        Network.FUSED_RECURRENCE = True
    assert torch.allclose(whole[1].float(), ref[1].float(), rtol=3e-2, atol=3e-2)
    assert torch.allclose(whole[2].float(), ref[2].float(), rtol=3e-2, atol=3e-2)


@pytest.mark.parametrize("B,T,N", [(6, 5, 7), (3, 16, 40), (4, 3, 48), (5, 2, 1), (4, 6, 16), (3, 4, 17), (3, 5, 32), (2, 3, 33), (3, 4, 24),  # <= 16 / <= 32: csrc/mapf_recur*_nt1.hip / _nt2.hip
                                   (3, 4, 64), (2, 3, 100), (2, 5, 128), (3, 2, 49)])  # > 48: csrc/mapf_recur_wide*.hip
def test_bptt_kernels_match_pytorch_recurrence(B, T, N):
    """mapf_recurrent_forward_save + mapf_recurrent_backward (the whole T-step GRU / CommBlock recurrence forward and
    backward in two launches) against the PyTorch-level recurrence at the same bf16 precision: Q-values and every
    parameter gradient of `bootstrap`."""
    from mapf_rl_amd.model import Network

    torch.manual_seed(B * 10 + N)
    net = Network().cuda()
    with torch.no_grad():
        for p in net.parameters():
            if p.dim() == 1:
                p.uniform_(-0.1, 0.1)
    g = torch.Generator(device="cuda").manual_seed(N + T)
    obs = (torch.rand((B, T, N, 6, 9, 9), device="cuda", generator=g) < 0.3).to(torch.bfloat16)
    steps = torch.randint(1, T + 1, (B,), device="cuda", generator=g)
    hidden = (torch.randn((B * N, 256), device="cuda", generator=g) * 0.3).to(torch.bfloat16)
    comm = torch.rand((B, T, N, N), device="cuda", generator=g) < 0.3
    comm |= torch.eye(N, dtype=torch.bool, device="cuda")
    if N > 2:
        comm[0, :, 1] = torch.eye(N, dtype=torch.bool, device="cuda")[1]
    comm[-1, -1] = False   # a zero-padded window row: all-False mask rows must neither produce NaN (SDPA path) nor gradients
    if N > 3:
        comm[-1, :, 2] = False   # a padded agent (smaller curriculum level): its mask row is all-False at every step
    steps[-1] = T
    r = torch.randn((B, 5), device="cuda", generator=g)
    res = {}
    for fused in (True, False):
        Network.FUSED_BPTT = fused
        try:
            net.zero_grad()
            q = net.bootstrap(obs, steps, hidden, comm)
            (q * r).sum().backward()
            res[fused] = (q.detach().clone(), {k: p.grad.detach().clone() for k, p in net.named_parameters()})
        finally:
            Network.FUSED_BPTT = True
    assert torch.allclose(res[True][0], res[False][0], rtol=3e-2, atol=3e-2)
    scale = max(float(b.norm()) for b in res[False][1].values())
    for k in res[True][1]:
        a, b = res[True][1][k], res[False][1][k]
        assert float((a - b).norm()) <= 8e-2 * float(b.norm()) + 1e-4 * scale, (k, float((a - b).norm()), float(b.norm()))


def test_policy_head_kernel_equals_the_module_statement():
    """mapf_q_head (dueling head + arg-max of the policy's forward, one launch) against model.py:216-220 in fp32 on the same bf16
    states: Q-values to 1e-5, the greedy action equal wherever the top-2 gap exceeds that."""
    from mapf_rl_amd.fused import q_head_infer
    from mapf_rl_amd.model import Network

    torch.manual_seed(5)
    net = Network().cuda().eval()
    for rows in (1, 7, 4096, 50001):
        h = (torch.randn((rows, 256), device="cuda") * 0.7).to(torch.bfloat16)
        q, act = q_head_infer(h, net.adv, net.state)
        with torch.no_grad():
            adv = torch.nn.functional.linear(h.float(), net.adv.weight, net.adv.bias)
            want = torch.nn.functional.linear(h.float(), net.state.weight, net.state.bias) + adv - adv.mean(-1, keepdim=True)
        assert q.shape == (rows, 5) and act.shape == (rows,) and act.dtype == torch.int64
        assert torch.allclose(q, want, rtol=1e-5, atol=1e-5), float((q - want).abs().max())
        top2 = want.topk(2, dim=-1).values
        clear = (top2[:, 0] - top2[:, 1]) > 1e-4
        assert torch.equal(act[clear], want.argmax(-1)[clear]) and torch.equal(act, q.argmax(-1))
    q_out, a_out = torch.empty((7, 5), device="cuda"), torch.empty(7, dtype=torch.int64, device="cuda")
    q2, a2 = q_head_infer(h[:7].contiguous(), net.adv, net.state, q_out, a_out)
    assert q2.data_ptr() == q_out.data_ptr() and a2.data_ptr() == a_out.data_ptr()


def test_cached_input_projection_weight_follows_the_parameter():
    """PackedRecurrence.input_weight: the bf16 copy of recurrent.weight_ih is converted once per parameter version, not per step; a
    changed parameter is picked up, in place (same address: a captured graph holds it) when asked to."""
    from mapf_rl_amd.fused import PackedRecurrence
    from mapf_rl_amd.model import Network

    torch.manual_seed(11)
    net = Network().cuda().eval()
    pk = PackedRecurrence()
    w0 = pk.input_weight(net)
    assert w0.dtype == torch.bfloat16 and torch.equal(w0, net.recurrent.weight_ih.detach().to(torch.bfloat16))
    assert pk.input_weight(net) is w0  # nothing changed: no conversion
    with torch.no_grad():
        net.recurrent.weight_ih.mul_(0.5)
    w1 = pk.input_weight(net, inplace=True)
    assert w1.data_ptr() == w0.data_ptr() and torch.equal(w1, net.recurrent.weight_ih.detach().to(torch.bfloat16))
    with torch.no_grad():
        net.recurrent.weight_ih.add_(1.0)
    w2 = pk.input_weight(net)  # not in place: a fresh buffer (a launch in flight may still read the old one)
    assert w2.data_ptr() != w1.data_ptr() and torch.equal(w2, net.recurrent.weight_ih.detach().to(torch.bfloat16))


@pytest.mark.parametrize("R", [1, 63, 64, 65, 1000, 40000])
def test_input_projection_kernel_for_all_rows_and_for_a_row_list(R):
    """mapf_input_proj_rows (csrc/mapf_inproj.hip; reference model.py:191, the W_ih x half of the GRUCell): every row against the fp32
    product of the same bf16 operands (one bf16 rounding of the result: 2^-8 relative); with a row list + device-side count only the
    listed rows are written, with the same bits as the all-rows launch; a count of zero writes nothing."""
    from mapf_rl_amd._lib import check, lib
    from mapf_rl_amd.fused import INPROJ_PACKED_ELEMS, _ptr, input_proj_rows

    torch.manual_seed(R)
    w = torch.randn(768, 784, device="cuda") * 0.05
    wp = torch.empty(INPROJ_PACKED_ELEMS, dtype=torch.bfloat16, device="cuda")
    check(lib.mapf_input_proj_pack(_ptr(w), _ptr(wp), None), "mapf_input_proj_pack")
    x = (torch.randn(R, 784, device="cuda") * 0.5).to(torch.bfloat16)
    want = x.float() @ w.to(torch.bfloat16).float().t()
    got = input_proj_rows(x, wp)
    assert got.shape == (R, 768) and got.dtype == torch.bfloat16
    assert torch.all((got.float() - want).abs() <= 2.0 ** -8 * want.abs() + 1e-6), float((got.float() - want).abs().max())
    n = max(1, R // 3)
    rows = torch.randperm(R, device="cuda")[:n].to(torch.int32).contiguous()
    cnt = torch.tensor([n], dtype=torch.int32, device="cuda")
    out = torch.full((R, 768), 7.0, dtype=torch.bfloat16, device="cuda")
    lst = torch.cat([rows, torch.full((5,), R - 1, dtype=torch.int32, device="cuda")])  # (entries behind the count are not read)
    input_proj_rows(x, wp, out, lst, cnt)
    sel = torch.zeros(R, dtype=torch.bool, device="cuda")
    sel[rows.long()] = True
    assert torch.equal(out[sel], got[sel]) and bool((out[~sel] == 7.0).all())
    out.fill_(7.0)
    input_proj_rows(x, wp, out, lst, torch.zeros(1, dtype=torch.int32, device="cuda"))
    assert bool((out == 7.0).all())


@pytest.mark.parametrize("B,S,T", [(6, 8, 3), (8, 4, 2), (4, 8, 18)])
def test_tiles_of_several_windows_step_like_one_window_per_environment(B, S, T):
    """mapf_recurrent_{infer,forward_save,backward}_packed with agent0_stride = S: 16 / S windows of S agents share a 16-row tile under a
    block-diagonal mask.  Row by row the same bits as one window per environment (plain entry points, E = B, N = S): hidden states,
    agent-0 states, every saved row tensor, every row gradient; the per-tile bias column sums add up to the per-window ones."""
    import ctypes

    from mapf_rl_amd import fused
    from mapf_rl_amd._lib import lib, check

    K, dev, bf = 16 // S, torch.device("cuda"), torch.bfloat16
    g = torch.Generator(device="cuda").manual_seed(B * 100 + S)
    w = (torch.randn(fused.RECUR_WEIGHT_ELEMS, device=dev, generator=g) * 0.05).to(bf)
    wt = (torch.randn(fused.RECUR_WEIGHT_ELEMS, device=dev, generator=g) * 0.05).to(bf)
    bias = torch.randn(fused.RECUR_BIAS_ELEMS, device=dev, generator=g) * 0.1
    gi = (torch.randn((T, B, S, 768), device=dev, generator=g) * 0.5).to(bf)
    h0 = (torch.randn((B, S, 256), device=dev, generator=g) * 0.3).to(bf)
    comm = (torch.rand((T, B, S, S), device=dev, generator=g) < 0.4) | torch.eye(S, dtype=torch.bool, device=dev)
    comm[:, 1, 2] = torch.eye(S, dtype=torch.bool, device=dev)[2]  # an agent without partners
    comm = comm.to(torch.uint8).contiguous()
    tiles = torch.zeros((T, B // K, 16, 16), dtype=torch.uint8, device=dev)
    for b in range(B):
        k = b % K
        tiles[:, b // K, k * S:(k + 1) * S, k * S:(k + 1) * S] = comm[:, b]
    d_a0 = (torch.randn((T, B, 256), device=dev, generator=g) * 0.1).to(bf)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    R = T * B * S

    def run(E, N, masks, stride):
        saves = [torch.zeros(s_, dtype=bf, device=dev) for s_ in ((R, 256), (R, 1024), (2, R, 256), (2, R, 384), (2, R, 128), (2, R, 64), (2, R, 1024),
                                                                 (2, T * B, 2, 48, 64))]
        outs = [torch.zeros(s_, dtype=bf, device=dev) for s_ in ((R, 768), (R, 768), (2, R, 768), (2, R, 768), (2, R, 64), (2, R, 384))]
        outs.append(torch.zeros((E, 2432), dtype=torch.float32, device=dev))
        h_out, a0 = torch.zeros((B, S, 256), dtype=bf, device=dev), torch.zeros((T, B, 256), dtype=bf, device=dev)
        h_inf, a_inf = torch.zeros_like(h_out), torch.zeros_like(a0)
        sp = (ctypes.c_void_p * 8)(*[t.data_ptr() for t in saves])
        op = (ctypes.c_void_p * 7)(*[t.data_ptr() for t in outs])
        p = lambda t: ctypes.c_void_p(t.data_ptr())
        check(lib.mapf_recurrent_infer_packed(p(gi), p(h0), p(masks), p(w), p(bias), T, E, N, p(h_inf), p(a_inf), None, 0, stride, st), "infer")
        check(lib.mapf_recurrent_forward_save_packed(p(gi), p(h0), p(masks), p(w), p(bias), T, E, N, p(h_out), p(a0), sp, None, 0, stride, st), "save")
        check(lib.mapf_recurrent_backward_packed(sp, p(masks), p(d_a0), p(wt), T, E, N, op, None, 0, stride, st), "bwd")
        torch.cuda.synchronize()
        return h_inf, a_inf, h_out, a0, saves[:7], outs

    plain = run(B, S, comm, 0)
    tiled = run(B // K, 16, tiles, S)
    for k in range(4):
        assert not plain[k].isnan().any()
        assert torch.equal(plain[k], tiled[k]), k
    assert torch.equal(plain[0], plain[2]) and torch.equal(plain[1], plain[3])  # (inference == training forward)
    for k, (a, b_) in enumerate(zip(plain[4], tiled[4])):
        assert torch.equal(a, b_), ("saved", k)
    for k in range(6):
        assert plain[5][k].abs().sum() > 0 and torch.equal(plain[5][k], tiled[5][k]), ("row gradient", k)
    assert torch.allclose(plain[5][6].sum(0), tiled[5][6].sum(0), rtol=1e-3, atol=1e-3)
