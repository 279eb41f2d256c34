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


# ---- deterministic network weights (identical on every box; independent of torch's RNG/version) ----
def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)).astype(np.uint64)
    z = x
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def det_state_dict(shapes, seed=1234):
    """shapes: ordered dict name -> shape.  Returns name -> float32 array, values uniform in +-bound with a
    Xavier-like bound for >=2-d tensors and +-0.05 for 1-d (biases), from a counter-based hash."""
    out = {}
    with np.errstate(over="ignore"):
        for p, (name, shape) in enumerate(shapes.items()):
            n = int(np.prod(shape))
            ctr = np.arange(n, dtype=np.uint64) + np.uint64(seed) * np.uint64(1 << 40) + np.uint64(p) * np.uint64(1 << 32)
            u = (_splitmix64(_splitmix64(ctr)) >> np.uint64(40)).astype(np.float64) / float(1 << 24)
            if len(shape) >= 2:
                fan_out = shape[0] * int(np.prod(shape[2:]))
                fan_in = shape[1] * int(np.prod(shape[2:]))
                bound = np.sqrt(6.0 / (fan_in + fan_out))
            else:
                bound = 0.05
            out[name] = ((u * 2.0 - 1.0) * bound).astype(np.float32).reshape(shape)
    return out


# ---- compact fingerprints of large gradient tensors (goldens hold these instead of 8 MB of raw gradients) ----
NPROJ = 24


def sign_matrix(n, k=NPROJ, seed=99):
    """k deterministic +-1 vectors of length n (counter-based hash, identical on every box): float32 [k, n]."""
    with np.errstate(over="ignore"):
        ctr = np.arange(n * k, dtype=np.uint64) + np.uint64(seed) * np.uint64(1 << 40)
        bits = (_splitmix64(_splitmix64(ctr)) >> np.uint64(63)).astype(np.float32)
    return (bits * 2.0 - 1.0).reshape(k, n)


def grad_fingerprint(g):
    """g: float array -> dict(norm, proj [NPROJ] = S g with S = sign_matrix, blocks = norms of up to 3 equal row blocks
    (the r/z/n gates of a GRU matrix or bias) ), all float64.  For an error e, E|S e| ~ ||e|| per projection."""
    g = np.asarray(g, np.float64)
    flat = g.reshape(-1)
    nb = 3 if (g.shape[0] % 3 == 0 and g.shape[0] >= 3) else 1
    blocks = np.array([np.linalg.norm(b) for b in np.split(flat, nb)])
    return dict(norm=np.linalg.norm(flat), proj=sign_matrix(flat.size).astype(np.float64) @ flat, blocks=blocks)
