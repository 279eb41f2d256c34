"""CPU, build container only: the oracle against the LIVE reference (imported via oracle/ref_harness.py)
on fresh seeded scenarios -- beyond what the committed goldens hold.  Skipped where /root/reference is
absent (e.g. the GPU box)."""
import random

import numpy as np
import pytest

from oracle import oracle, ref_harness as rh
from tests import helpers as H

pytestmark = [pytest.mark.needs_reference,
              pytest.mark.skipif(not rh.reference_available(), reason="reference not present")]


@pytest.fixture(scope="module")
def ref():
    return rh.load_reference()


@pytest.mark.parametrize("L,N,seed", [(8, 10, 1), (10, 30, 2), (16, 40, 3), (20, 6, 4), (32, 40, 5)])
def test_random_scenarios_live(ref, L, N, seed):
    maps, agents, goals = H.random_scenarios(1, L, N, 0.3 if N * 4 < L * L else 0.1, seed)
    m, a, g = maps[0], agents[0], goals[0]
    env = ref.environment.Environment()
    env.load(m.astype(np.int64), a.astype(np.int64), g.astype(np.int64))
    nv = oracle.navi(m, g)
    r = H.R
    assert np.array_equal(env.navi_map[:, :, r:-r, r:-r].astype(np.uint8), nv)
    rng = random.Random(seed)
    ag = a.copy()
    obs, _ = env.observe()
    for t in range(60):
        if t % 2:
            acts = [rng.randrange(5) for _ in range(N)]
        else:  # contention: follow the heuristic
            acts = []
            for i in range(N):
                fl = np.nonzero(obs[i, 2:6, r, r])[0]
                acts.append(0 if len(fl) == 0 else 1 + int(fl[rng.randrange(len(fl))]))
        (obs, pos), rew, done, _ = env.step(acts)
        st, ag, rc, dn = oracle.step(m, ag, g, acts)
        assert st == 0 and np.array_equal(ag, pos) and H.rewards_match(rc, rew) and dn == done
        assert np.array_equal(oracle.observe(m, ag, nv), obs.astype(np.uint8))


def test_bad_action_is_assertion(ref):
    env = ref.environment.Environment()
    m = np.zeros((5, 5), np.int64)
    env.load(m, np.array([[0, 0]]), np.array([[1, 1]]))
    with pytest.raises(AssertionError):
        env.step([5])
