"""CPU: the host scenario generator (mapf_generate) against statistics captured from the REFERENCE's own generator
(reference environment.py:21-70,100-138; tests/golden/gen_stats.npz, made by tests/golden/make_gen_stats.py): realised
obstacle density under triangular(0, 0.33, 0.5), component structure, start-goal BFS and Manhattan distance, share of
agents placed in the largest component, failure / redraw rate.  Two-sample KS <= 0.05 (tests/gen_stats.py)."""
import numpy as np
import pytest

import mapf_rl_amd as M
from tests import gen_stats as GS


@pytest.mark.parametrize("N,L,E", [(6, 20, 6000), (40, 32, 4000), (16, 40, 4000)])
def test_generator_matches_reference_statistics(N, L, E):
    maps, agents, goals, redraws = M.generate_scenarios(E, L, N, -1.0, seed=11)
    res = GS.check(N, L, GS.batch_stats(maps, agents, goals))
    # the reference raises ValueError when placement runs out of cells (environment.py:120), this generator redraws: both rare
    assert res["ref_failure_rate"] <= 0.01 and redraws / E <= 0.01, (res, redraws)
