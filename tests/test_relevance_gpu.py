"""mapf_window_relevance (csrc/mapf_dqn.hip) against the PyTorch restatement of `relevance()`, and `Network.bootstrap` with
only the reachable observations encoded against the same call encoding all of them: the Q-values must be the same bits (the
encoder is per observation, the pruned agents are never read by an agent that matters), the parameter gradients the same up to
the order of the weight-/bias-gradient partial sums."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _torch_relevance(comm, steps):
    B, T, N, _ = comm.shape
    last = steps.view(B) - 1
    a0 = torch.zeros((B, N), dtype=torch.bool, device=comm.device)
    a0[:, 0] = True
    r = torch.zeros_like(a0)
    rel = torch.zeros((T, B, N), dtype=torch.bool, device=comm.device)
    for t in range(T - 1, -1, -1):
        r = r | (a0 & (last == t).view(B, 1))
        for _ in range(2):
            r = r | (r.unsqueeze(2) & comm[:, t]).any(dim=1)
        rel[t] = r
    return rel


@pytest.mark.parametrize("B,T,N,p", [(192, 18, 40, 0.04), (7, 16, 128, 0.01), (33, 18, 6, 0.2), (5, 3, 1, 1.0), (64, 18, 49, 0.0)])
def test_kernel_equals_restatement(B, T, N, p):
    from mapf_rl_amd.model import relevance

    g = torch.Generator(device="cuda").manual_seed(B + N)
    comm = (torch.rand((B, T, N, N), device="cuda", generator=g) < p) | torch.eye(N, dtype=torch.bool, device="cuda")
    steps = torch.randint(1, T + 1, (B,), device="cuda", generator=g)
    assert torch.equal(relevance(comm, steps), _torch_relevance(comm, steps))


def test_argument_checks():
    from mapf_rl_amd._lib import ERR_INVALID_ARG, lib

    comm = torch.ones((1, 2, 3, 3), dtype=torch.uint8, device="cuda")
    steps = torch.ones(1, dtype=torch.int64, device="cuda")
    rel = torch.empty((2, 1, 3), dtype=torch.uint8, device="cuda")
    assert lib.mapf_window_relevance(comm.data_ptr(), steps.data_ptr(), 2, 1, 129, rel.data_ptr(), None) == ERR_INVALID_ARG
    assert lib.mapf_window_relevance(None, steps.data_ptr(), 2, 1, 3, rel.data_ptr(), None) == ERR_INVALID_ARG
    assert lib.mapf_window_relevance(comm.data_ptr(), steps.data_ptr(), 0, 1, 3, rel.data_ptr(), None) == ERR_INVALID_ARG


@pytest.mark.parametrize("N", [6, 40, 64, 128])
def test_pruned_bootstrap_is_the_same_function(N):
    from mapf_rl_amd.model import Network, comm_mask_from_pos

    torch.manual_seed(N)
    net = Network().cuda()
    g = torch.Generator(device="cuda").manual_seed(7)
    B, T = 24, 18
    obs = (torch.rand((B, T, N, 6, 9, 9), device="cuda", generator=g) < 0.3).to(torch.bfloat16)
    pos = torch.randint(0, 24, (B * T, N, 2), device="cuda", generator=g).float()
    comm = comm_mask_from_pos(pos).view(B, T, N, N)
    steps = torch.randint(1, T + 1, (B,), device="cuda", generator=g)
    hidden = (torch.randn((B * N, 256), device="cuda", generator=g) * 0.3)
    out = {}
    try:
        for prune in (False, True):
            Network.PRUNE_UNREACHABLE = prune
            net.zero_grad()
            with torch.no_grad():
                q_eval = net.bootstrap(obs, steps, hidden, comm)
            q = net.bootstrap(obs, steps, hidden, comm)
            (q * torch.arange(1, 6, device="cuda")).sum().backward()
            out[prune] = (q_eval, q.detach(), {k: p.grad.clone() for k, p in net.named_parameters()})
    finally:
        Network.PRUNE_UNREACHABLE = True
    if N <= 48:
        assert torch.equal(out[False][0], out[True][0]) and torch.equal(out[False][1], out[True][1])
    else:  # the reachable agents run through the 48-agent kernels in a different order (softmax sums): bf16 tolerance instead of bits
        for a, b_ in ((out[False][0], out[True][0]), (out[False][1], out[True][1])):
            assert bool(((a - b_).abs() <= 2e-2 * torch.clamp(a.abs(), min=1.0)).all()), float((a - b_).abs().max())
    tot = torch.sqrt(sum((g_ ** 2).sum() for g_ in out[False][2].values()))
    for k, g0 in out[False][2].items():
        d = float((g0 - out[True][2][k]).norm())
        assert d <= (2e-3 if N <= 48 else 3e-2) * float(g0.norm()) + (1e-5 if N <= 48 else 1e-3) * float(tot), (k, d, float(g0.norm()))
