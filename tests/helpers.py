"""Shared helpers for the tests: golden-vector access and seeded scenario/tape generation (numpy only)."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
REWARD_VALUES = np.array([-0.075, 0.0, -0.075, -0.5, 3.0])  # reference config.py:8-12 in MAPF_RC_* order
R = 4


def load_npz(name):
    return np.load(os.path.join(GOLDEN, name))


def unpack_bits(packed, shape):
    n = int(np.prod(shape))
    return np.unpackbits(packed, bitorder="little")[:n].reshape(shape)


def fixture_cases(z):
    """Yields (prefix, map, agents, goals) for the pkl-derived golden cases."""
    for nag in (16, 32, 64):
        for c in range(3):
            pre = "fix%d_c%d_" % (nag, c)
            yield pre, z[pre + "map"], z[pre + "agents"].astype(np.int16), z[pre + "goals"].astype(np.int16)


def dense_cases(z):
    for i in range(int(z["num_cases"])):
        pre = "dense%d_" % i
        yield pre, z[pre + "map"], z[pre + "agents"].astype(np.int16), z[pre + "goals"].astype(np.int16)


POLICIES = ("uniform", "mixed", "greedy")


def rewards_match(rclass, ref_rewards):
    """reference rewards are python floats from config.reward_fn; classes map onto them exactly."""
    return np.array_equal(REWARD_VALUES[np.asarray(rclass)], np.asarray(ref_rewards))


def random_scenarios(E, L, N, density, seed):
    """Test-side scenario sampler (NOT the product generator): Bernoulli map, distinct free cells for
    agents and goals.  No connectivity guarantee -- unreachable goals are a legal, interesting input."""
    rng = np.random.RandomState(seed)
    maps = np.zeros((E, L, L), np.int8)
    agents = np.zeros((E, N, 2), np.int16)
    goals = np.zeros((E, N, 2), np.int16)
    for e in range(E):
        for attempt in range(1000):
            m = (rng.random_sample((L, L)) < density).astype(np.int8)
            free = np.argwhere(m == 0)
            if len(free) >= 2 * N:
                break
        else:
            raise ValueError("random_scenarios: %dx%d at density %.2f cannot host %d agents" % (L, L, density, N))
        pick = free[rng.permutation(len(free))[:2 * N]]
        maps[e], agents[e], goals[e] = m, pick[:N], pick[N:]
    return maps, agents, goals


def random_tape(T, E, N, seed, p_stay=0.2):
    rng = np.random.RandomState(seed)
    tape = rng.randint(0, 5, size=(T, E, N)).astype(np.int8)
    return tape
