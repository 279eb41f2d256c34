"""CPU: the host scenario generator (mapf_generate; reference rule environment.py:100-138) --
invariants the reference guarantees by construction, and distribution sanity."""
import numpy as np
import pytest

import mapf_rl_amd as M
from oracle import oracle


@pytest.mark.parametrize("L,N,rho", [(32, 40, 0.3), (16, 40, 0.3), (10, 1, -1.0), (40, 16, 0.3), (20, 6, -1.0)])
def test_invariants(L, N, rho):
    E = 24
    maps, agents, goals, redraws = M.generate_scenarios(E, L, N, rho, seed=7)
    assert set(np.unique(maps)) <= {0, 1}
    for e in range(E):
        cells = np.concatenate([agents[e], goals[e]])
        assert cells.min() >= 0 and cells.max() < L
        # all 2N cells distinct and free (environment.py:129-135: removed from the partition once used)
        keys = cells[:, 0] * L + cells[:, 1]
        assert len(np.unique(keys)) == 2 * N
        assert maps[e][cells[:, 0], cells[:, 1]].sum() == 0
        # start and goal of an agent are in the same connected component: goal reachable
        for i in range(min(N, 8)):
            d = oracle.dist(maps[e], goals[e, i])
            assert d[agents[e, i, 0], agents[e, i, 1]] < 2147483647


def test_deterministic_and_seed_sensitive():
    a = M.generate_scenarios(4, 16, 8, 0.3, seed=1)
    b = M.generate_scenarios(4, 16, 8, 0.3, seed=1)
    c = M.generate_scenarios(4, 16, 8, 0.3, seed=2)
    assert all(np.array_equal(x, y) for x, y in zip(a[:3], b[:3]))
    assert not np.array_equal(a[0], c[0])


def test_density_distribution():
    maps, _, _, _ = M.generate_scenarios(256, 32, 4, 0.3, seed=3)
    assert abs(maps.mean() - 0.3) < 0.01
    # triangular(0, 0.33, 0.5): mean (0+0.33+0.5)/3
    maps, _, _, _ = M.generate_scenarios(512, 20, 2, -1.0, seed=4)
    assert abs(maps.mean() - (0.33 + 0.5) / 3) < 0.02
    assert maps.reshape(512, -1).mean(1).max() < 0.62


def test_no_space_is_value_error():
    with pytest.raises(ValueError):
        M.generate_scenarios(1, 4, 40, 0.3, seed=0)  # 16 cells cannot host 80 distinct positions
