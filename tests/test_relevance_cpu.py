"""`relevance()` (mapf_rl_amd/model.py): the (step, window, agent) entries of a training batch that can reach agent 0's Q-value.
Soundness against the reference-shaped module path on CPU: change one observation at a time -- wherever the Q-value of a
window moves, the entry must have been marked (reference model.py:242-262: the only cross-agent path is CommBlock's mask)."""
import torch

from mapf_rl_amd.model import Network, relevance


def _loop(comm_mask, steps):
    """independent restatement: explicit sets, one step at a time"""
    B, T, N, _ = comm_mask.shape
    out = torch.zeros((T, B, N), dtype=torch.bool)
    for b in range(B):
        need = set()
        for t in range(T - 1, -1, -1):
            if t == int(steps[b]) - 1:
                need.add(0)
            for _ in range(2):
                need |= {j for i in need for j in range(N) if comm_mask[b, t, i, j]}
            for j in need:
                out[t, b, j] = True
    return out


def test_relevance_is_the_backward_closure_of_agent_zero():
    g = torch.Generator().manual_seed(0)
    for B, T, N, p in ((4, 6, 7, 0.15), (3, 18, 12, 0.1), (2, 5, 1, 0.5), (5, 4, 9, 0.0)):
        comm = (torch.rand((B, T, N, N), generator=g) < p) | torch.eye(N, dtype=torch.bool)
        steps = torch.randint(1, T + 1, (B,), generator=g)
        rel = relevance(comm, steps)
        assert rel.shape == (T, B, N) and rel.dtype == torch.bool
        assert torch.equal(rel, _loop(comm, steps))
        for b in range(B):
            assert bool(rel[int(steps[b]) - 1, b, 0]) and not bool(rel[int(steps[b]):, b].any())


def test_every_observation_that_moves_q_is_marked():
    torch.manual_seed(1)
    net = Network().eval()
    g = torch.Generator().manual_seed(2)
    B, T, N = 3, 5, 5
    obs = (torch.rand((B, T, N, 6, 9, 9), generator=g) < 0.3).float()
    comm = (torch.rand((B, T, N, N), generator=g) < 0.2) | torch.eye(N, dtype=torch.bool)
    steps = torch.tensor([5, 3, 1])
    hidden = torch.randn((B * N, 256), generator=g) * 0.3
    rel = relevance(comm, steps)
    with torch.no_grad():
        q0 = net.bootstrap(obs, steps, hidden, comm)
        moved = torch.zeros((T, B, N), dtype=torch.bool)
        for b in range(B):
            for t in range(T):
                for n in range(N):
                    o2 = obs.clone()
                    o2[b, t, n] = 1.0 - o2[b, t, n]
                    q = net.bootstrap(o2, steps, hidden, comm)
                    assert torch.equal(q[torch.arange(B) != b], q0[torch.arange(B) != b])  # windows are independent
                    moved[t, b, n] = not torch.equal(q[b], q0[b])
    assert not bool((moved & ~rel).any()), (moved & ~rel).nonzero()
    assert bool(moved.any()) and float(rel.float().mean()) < 0.9  # (the test would be empty if nothing moved / everything were marked)
