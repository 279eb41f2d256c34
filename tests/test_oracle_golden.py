"""CPU: pins the oracle (oracle/mapf_oracle.c) against golden vectors captured from the unmodified
reference (tests/golden/make_goldens.py).  The oracle is then the checker for the HIP path."""
import hashlib

import numpy as np
import pytest

from oracle import oracle
from tests import helpers as H


def _check_tapes(z, pre, m, a, g):
    nv = oracle.navi(m, g)
    N = a.shape[0]
    for pol in H.POLICIES:
        acts = z[pre + pol + "_actions"]
        pos = z[pre + pol + "_pos"]
        rew = z[pre + pol + "_rewards"]
        done = z[pre + pol + "_done"]
        sha = z[pre + pol + "_obs_sha"]
        bits = z[pre + pol + "_obs_bits"]
        assert np.array_equal(pos[0], a)
        ag = a.copy()
        ob = oracle.observe(m, ag, nv)
        assert hashlib.sha256(ob.tobytes()).digest() == sha[0].tobytes()
        for t in range(acts.shape[0]):
            st, ag, rc, dn = oracle.step(m, ag, g, acts[t])
            assert st == 0
            assert np.array_equal(ag, pos[t + 1]), (pre, pol, t)
            assert H.rewards_match(rc, rew[t]), (pre, pol, t)
            assert dn == bool(done[t])
            ob = oracle.observe(m, ag, nv)
            assert hashlib.sha256(ob.tobytes()).digest() == sha[t + 1].tobytes(), (pre, pol, t)
            if t + 1 < bits.shape[0]:
                assert np.array_equal(ob, H.unpack_bits(bits[t + 1], (N, 6, 9, 9)))


def test_fixture_navi_and_dist():
    z = H.load_npz("env_fixtures.npz")
    for pre, m, a, g in H.fixture_cases(z):
        N, L = a.shape[0], m.shape[0]
        nv = oracle.navi(m, g)
        assert np.array_equal(nv, H.unpack_bits(z[pre + "navi_bits"], (N, 4, L, L))), pre
        # search.compute_heuristics (search.py:24-55) == BFS distance of get_navi_map
        assert np.array_equal(oracle.dist(m, g[0]), z[pre + "dist0"]), pre


def test_fixture_trajectories():
    z = H.load_npz("env_fixtures.npz")
    for pre, m, a, g in H.fixture_cases(z):
        _check_tapes(z, pre, m, a, g)


def test_dense_trajectories():
    z = H.load_npz("env_dense.npz")
    for pre, m, a, g in H.dense_cases(z):
        N, L = a.shape[0], m.shape[0]
        assert np.array_equal(oracle.navi(m, g), H.unpack_bits(z[pre + "navi_bits"], (N, 4, L, L)))
        _check_tapes(z, pre, m, a, g)


# hand-derived answers, independent of the reference run (pos after the step, reward classes)
KNOWN = {
    "K1_swap": ([[5, 0], [2, 2], [2, 3]], [2, 3, 3], False),
    "K2_swap_agent0": ([[2, 2], [2, 3]], [3, 3], False),
    "K3_rotation": ([[1, 2], [2, 2], [2, 1], [1, 1]], [0, 0, 0, 0], False),
    "K4_follow": ([[2, 2], [2, 3], [2, 4]], [0, 0, 0], False),
    "K5_chain_obstacle": ([[2, 3], [2, 2], [2, 1], [2, 0]], [3, 3, 3, 3], False),
    "K6_vertex_lowest_id": ([[2, 2], [2, 3], [1, 2]], [0, 3, 3], False),
    "K7_into_stationary": ([[2, 2], [2, 3]], [3, 2], False),
    "K8_winner_evicted": ([[1, 3], [3, 3], [2, 3]], [3, 3, 3], False),
    "K8b_rotation_broken_by_lower_id": ([[0, 1], [1, 1], [1, 2], [2, 2], [2, 1]], [3] * 5, False),
    "K8c_rotation_higher_id_outsider": ([[1, 2], [2, 2], [2, 1], [1, 1], [0, 1]], [0, 0, 0, 0, 3], False),
    "K9_oob": ([[0, 2], [4, 1], [2, 0], [3, 4]], [3, 3, 3, 3], False),
    "K10_obstacle": ([[1, 0], [0, 1]], [3, 3], False),
    "K11_finish": ([[1, 2], [3, 3]], [4, 4], True),
    "K12_stay_move": ([[0, 0], [2, 2], [3, 4]], [1, 2, 0], False),
    "K13_cascade_behind_loser": ([[2, 2], [2, 3], [2, 4], [2, 5]], [0, 3, 3, 3], False),
}


def test_known_answers():
    z = H.load_npz("env_known.npz")
    names = [str(n) for n in z["names"]]
    assert set(KNOWN) <= set(names)
    for name in names:
        m, a, g = z[name + "_map"], z[name + "_agents"].astype(np.int16), z[name + "_goals"].astype(np.int16)
        st, ag, rc, dn = oracle.step(m, a, g, z[name + "_actions"])
        assert st == 0
        assert np.array_equal(ag, z[name + "_pos"]), name
        assert H.rewards_match(rc, z[name + "_rewards"]), name
        assert dn == bool(z[name + "_done"]), name
        ob = oracle.observe(m, ag, oracle.navi(m, g))
        assert np.array_equal(ob, H.unpack_bits(z[name + "_obs_bits"], ob.shape)), name
        if name in KNOWN:
            pos, rcs, done = KNOWN[name]
            assert ag.tolist() == pos and rc.tolist() == rcs and dn == done, name


def test_error_codes():
    m = np.zeros((4, 4), np.int8)
    a = np.array([[0, 0], [1, 1]], np.int16)
    st, ag, _, _ = oracle.step(m, a, a, [5, 0])
    assert st == oracle.ERR_ACTION and np.array_equal(ag, a)
    st, _, _, _ = oracle.step(m, a, a, [-1, 0])
    assert st == oracle.ERR_ACTION


def test_rollout_matches_single_steps():
    maps, agents, goals = H.random_scenarios(6, 12, 20, 0.25, seed=3)
    tape = H.random_tape(20, 6, 20, seed=4)
    nv = oracle.navi_batch(maps, goals)
    out = oracle.rollout(maps, agents, goals, nv, tape, want_obs_last=True, want_hash=True, nthreads=2)
    assert out["status"] == 0
    for e in range(6):
        ag = agents[e].copy()
        for t in range(20):
            st, ag, rc, dn = oracle.step(maps[e], ag, goals[e], tape[t, e])
            assert np.array_equal(ag, out["pos"][t, e]) and np.array_equal(rc, out["rclass"][t, e])
            assert dn == bool(out["done"][t, e])
        assert np.array_equal(oracle.observe(maps[e], ag, nv[e]), out["obs_last"][e])
