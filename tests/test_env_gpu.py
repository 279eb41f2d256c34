"""GPU parity tests: the HIP path (through the C ABI, via mapf_rl_amd.VecEnvironment / Environment)
against (1) golden vectors captured from the unmodified reference and (2) the CPU oracle on seeded
inputs.  Everything here is integer/bit work: the bar is bit-exact."""
import hashlib

import numpy as np
import pytest
import torch

from oracle import oracle
from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def M():
    import mapf_rl_amd

    assert torch.cuda.is_available(), "-m gpu tests need a HIP device"
    return mapf_rl_amd


def _np(t):
    return t.cpu().numpy()


def _run_golden_tapes(M, z, pre, m, a, g):
    N, L = a.shape[0], m.shape[0]
    env = M.VecEnvironment(1, L, N)
    for pol in H.POLICIES:
        env.load(m[None], a[None], g[None])
        acts = z[pre + pol + "_actions"]
        pos = z[pre + pol + "_pos"]
        rew = z[pre + pol + "_rewards"]
        done = z[pre + pol + "_done"]
        sha = z[pre + pol + "_obs_sha"]
        bits = z[pre + pol + "_obs_bits"]
        obs, p = env.observe()
        assert np.array_equal(_np(p)[0], pos[0])
        assert hashlib.sha256(_np(obs)[0].tobytes()).digest() == sha[0].tobytes(), (pre, pol, "obs0")
        for t in range(acts.shape[0]):
            obs, p, r, d, rc = env.step(torch.from_numpy(acts[t][None]).cuda())
            assert np.array_equal(_np(p)[0], pos[t + 1]), (pre, pol, t)
            assert H.rewards_match(_np(rc)[0], rew[t]), (pre, pol, t)
            assert np.array_equal(_np(r)[0], rew[t].astype(np.float32)), (pre, pol, t)
            assert bool(_np(d)[0]) == bool(done[t])
            o = _np(obs)[0]
            assert hashlib.sha256(o.tobytes()).digest() == sha[t + 1].tobytes(), (pre, pol, t)
            if t + 1 < bits.shape[0]:
                assert np.array_equal(o, H.unpack_bits(bits[t + 1], (N, 6, 9, 9)))
        env.check_status()
        assert int(_np(env.steps())[0]) == acts.shape[0]


def test_golden_fixture_navi(M):
    z = H.load_npz("env_fixtures.npz")
    for pre, m, a, g in H.fixture_cases(z):
        N, L = a.shape[0], m.shape[0]
        env = M.VecEnvironment(1, L, N)
        env.load(m[None], a[None], g[None])
        assert np.array_equal(_np(env.navi_map())[0], H.unpack_bits(z[pre + "navi_bits"], (N, 4, L, L))), pre
        assert np.array_equal(_np(env.maps())[0], m)
        assert np.array_equal(_np(env.goals_pos())[0], g)


def test_golden_fixture_trajectories(M):
    z = H.load_npz("env_fixtures.npz")
    for pre, m, a, g in H.fixture_cases(z):
        _run_golden_tapes(M, z, pre, m, a, g)


def test_golden_dense_trajectories(M):
    z = H.load_npz("env_dense.npz")
    for pre, m, a, g in H.dense_cases(z):
        N, L = a.shape[0], m.shape[0]
        env = M.VecEnvironment(1, L, N)
        env.load(m[None], a[None], g[None])
        assert np.array_equal(_np(env.navi_map())[0], H.unpack_bits(z[pre + "navi_bits"], (N, 4, L, L))), pre
        _run_golden_tapes(M, z, pre, m, a, g)


def test_golden_known_answers(M):
    from tests.test_oracle_golden import KNOWN

    z = H.load_npz("env_known.npz")
    for name in [str(n) for n in z["names"]]:
        m, a, g = z[name + "_map"], z[name + "_agents"].astype(np.int16), z[name + "_goals"].astype(np.int16)
        env = M.VecEnvironment(1, m.shape[0], a.shape[0])
        env.load(m[None], a[None], g[None])
        obs, p, r, d, rc = env.step(torch.from_numpy(z[name + "_actions"][None]).cuda())
        env.check_status()
        assert np.array_equal(_np(p)[0], z[name + "_pos"]), name
        assert H.rewards_match(_np(rc)[0], z[name + "_rewards"]), name
        assert bool(_np(d)[0]) == bool(z[name + "_done"]), name
        assert np.array_equal(_np(obs)[0], H.unpack_bits(z[name + "_obs_bits"], tuple(obs.shape[1:]))), name
        if name in KNOWN:
            pos, rcs, done = KNOWN[name]
            assert _np(p)[0].tolist() == pos and _np(rc)[0].tolist() == rcs and bool(_np(d)[0]) == done, name


def _heuristic_tape_step(obs, rng, p_follow):
    """numpy policy on a batch: follow a navi flag of the own cell with prob p_follow, else uniform."""
    E, N = obs.shape[:2]
    flags = obs[:, :, 2:6, H.R, H.R].astype(bool)  # [E,N,4]
    score = rng.random_sample((E, N, 4)) * flags
    follow = np.where(flags.any(-1), 1 + score.argmax(-1), 0)
    uni = rng.randint(0, 5, size=(E, N))
    return np.where(rng.random_sample((E, N)) < p_follow, follow, uni).astype(np.int8)


@pytest.mark.parametrize("E,L,N,rho,T", [
    (64, 8, 10, 0.15, 40),     # tiny, dense in agents
    (64, 16, 40, 0.2, 40),     # BASELINE config 1 literal reading
    (32, 40, 16, 0.3, 40),     # fixture shape (W = uint64 rows)
    (128, 32, 40, 0.3, 64),    # BASELINE config 2 shape
    (16, 64, 40, 0.3, 32),     # config 3
    (8, 64, 128, 0.3, 32),     # config 5 shape (N > one wavefront)
    (16, 33, 7, 0.3, 24),      # odd sizes: L just over 32, N odd -> 4/1-byte store path
    (16, 10, 1, 0.2, 24),      # single agent
    (16, 12, 3, 0.2, 24),
    (16, 12, 6, 0.2, 24),      # reference default num_agents
    (4, 24, 200, 0.1, 16),     # N > 128
])
def test_differential_vs_oracle(M, E, L, N, rho, T):
    _differential_vs_oracle(M, E, L, N, rho, T)


# The launch shapes the automatic rules pick only at sizes no oracle comparison reaches in seconds (csrc/mapf_env.hip:
# step_nt_store -- non-temporal observation stores from 176 MB of observations per launch on; step_block_threads -- four waves per
# environment from 12,288 environments on; step_use_plane) -- forced through the handle's tuning overrides, which are read at
# creation time, so that EVERY step, position, reward class, done flag and the final observation goes against the sequential oracle.
@pytest.mark.parametrize("tune", [
    {"MAPF_STEP_NT": "1"},
    {"MAPF_STEP_THREADS": "256"},
    {"MAPF_STEP_NT": "1", "MAPF_STEP_THREADS": "256"},      # what frac_out_of_cache runs (bench.py)
    {"MAPF_STEP_NT": "1", "MAPF_STEP_THREADS": "256", "MAPF_STEP_PLANE": "0"},
    {"MAPF_STEP_PLANE": "0"},
    {"MAPF_STEP_PLANE": "1"},
    {"MAPF_STEP_THREADS": "64"},
], ids=lambda d: ",".join("%s=%s" % (k[10:], v) for k, v in d.items()))
@pytest.mark.parametrize("E,L,N,rho,T", [(128, 32, 40, 0.3, 64), (48, 20, 6, 0.2, 40), (16, 64, 40, 0.3, 24)])
def test_forced_launch_families_vs_oracle(M, tune, E, L, N, rho, T):
    _differential_vs_oracle(M, E, L, N, rho, T, tune=tune, every_obs=True)


def _differential_vs_oracle(M, E, L, N, rho, T, tune=None, every_obs=False):
    import os

    maps, agents, goals = H.random_scenarios(E, L, N, rho, seed=E * 1000 + L * 10 + N)
    os.environ.update(tune or {})
    try:
        env = M.VecEnvironment(E, L, N)
    finally:
        for k in (tune or {}):
            del os.environ[k]
    env.load(maps, agents, goals)
    nv = oracle.navi_batch(maps, goals)
    assert np.array_equal(_np(env.navi_map()), nv)
    rng = np.random.RandomState(N + L)
    obs, pos = env.observe()
    assert np.array_equal(_np(pos), agents)
    o0 = _np(obs)
    for e in range(min(E, 4)):
        assert np.array_equal(o0[e], oracle.observe(maps[e], agents[e], nv[e]))
    tape = np.zeros((T, E, N), np.int8)
    got_pos = np.zeros((T, E, N, 2), np.int16)
    got_rc = np.zeros((T, E, N), np.int8)
    got_done = np.zeros((T, E), np.uint8)
    got_hash = []
    for t in range(T):
        p_follow = (0.0, 0.8, 1.0)[t % 3]
        tape[t] = _heuristic_tape_step(_np(obs), rng, p_follow)
        obs, pos, rew, done, rc = env.step(torch.from_numpy(tape[t]).cuda())
        got_pos[t], got_rc[t], got_done[t] = _np(pos), _np(rc), _np(done)
        assert np.array_equal(_np(rew), H.REWARD_VALUES.astype(np.float32)[got_rc[t]])
        if every_obs:  # the observation of EVERY step (the stores are what the forced variants change), two environments each
            for e in (0, E - 1):
                assert np.array_equal(_np(obs[e]), oracle.observe(maps[e], got_pos[t, e], nv[e])), (t, e)
    env.check_status()
    ref = oracle.rollout(maps, agents, goals, nv, tape, want_obs_last=True)
    assert ref["status"] == 0
    assert np.array_equal(got_pos, ref["pos"])
    assert np.array_equal(got_rc, ref["rclass"])
    assert np.array_equal(got_done, ref["done"])
    assert np.array_equal(_np(obs), ref["obs_last"])
    assert np.array_equal(_np(env.steps()), np.full(E, T, np.int32))


@pytest.mark.parametrize("E,L,N", [(4096, 20, 6), (48, 10, 1), (8192, 15, 3), (24, 12, 2), (4096, 16, 4), (8192, 9, 5), (8, 32, 8), (8192, 10, 7)])
def test_packed_blocks_equal_one_block_per_environment(M, E, L, N):
    """Few agents: one workgroup steps G consecutive environments (env_step_kernel<..., G>, csrc/mapf_env.hip).  Same scenario and
    tape through the packed launch and through one block per environment (MAPF_STEP_GROUP=1): every output of every step is
    identical -- byte observations, bit-packed observation rows, positions, reward classes, rewards, done (environments finish at
    different steps), step counters -- and the trajectory equals the sequential oracle's."""
    import os

    maps, agents, goals, _ = M.generate_scenarios(E, L, N, 0.15, seed=E + L + N)
    envs = []
    for cap in ("8", "1"):
        os.environ["MAPF_STEP_GROUP"] = cap
        try:
            env = M.VecEnvironment(E, L, N)
        finally:
            del os.environ["MAPF_STEP_GROUP"]
        env.load(maps, agents, goals)
        envs.append(env)
    RD = envs[0].obs_bits_row_dwords
    bits = [torch.zeros((E, RD), dtype=torch.int32, device="cuda") for _ in envs]
    rng = np.random.RandomState(E * 7 + N)
    obs = [env.observe(obs_bits_out=b)[0] for env, b in zip(envs, bits)]
    assert torch.equal(obs[0], obs[1]) and torch.equal(bits[0], bits[1])
    T = 3 * L
    tape = np.zeros((T, E, N), np.int8)
    any_done = False
    for t in range(T):
        tape[t] = _heuristic_tape_step(_np(obs[0]), rng, (0.6, 1.0, 1.0)[t % 3])
        a = torch.from_numpy(tape[t]).cuda()
        outs = [env.step(a, obs_bits_out=b) for env, b in zip(envs, bits)]
        for x, y in zip(outs[0], outs[1]):
            assert torch.equal(x, y), t
        assert torch.equal(bits[0], bits[1]), t
        obs = [o[0] for o in outs]
        any_done |= bool(outs[0][3].any())
        # the bit-packed row is the byte observation, bit b of row e = byte b of environment e's block
        raw = np.unpackbits(_np(bits[0]).view(np.uint32).view(np.uint8).reshape(E, -1), axis=1, bitorder="little")
        assert np.array_equal(raw[:, :N * 486], _np(obs[0]).reshape(E, -1))
        assert not raw[:, N * 486:].any()
    for env in envs:
        env.check_status()
    assert any_done  # some environment reached its goals within 3 L greedy-ish steps
    nv = oracle.navi_batch(maps, goals)
    ref = oracle.rollout(maps, agents, goals, nv, tape, want_obs_last=True)
    assert np.array_equal(_np(envs[0].agents_pos()), ref["pos"][-1]) and np.array_equal(_np(obs[0]), ref["obs_last"])
    assert np.array_equal(_np(envs[0].steps()), np.full(E, T, np.int32))


@pytest.mark.parametrize("E,L,N,T,stride", [(4096, 32, 40, 24, 64), (4096, 64, 40, 16, 128), (2048, 64, 128, 10, 128),
                                            (16384, 32, 40, 12, 256), (32768, 32, 40, 8, 512)],
                         ids=["config2", "config3", "config5", "config2_x4_nt_stores_4_waves", "config2_x8_hbm_proper"])
def test_full_size_config_properties(M, E, L, N, T, stride):
    """BASELINE configs 2, 3 and 5 (per GPU) at FULL size -- 4096 x 32x32 x 40 agents, 4096 x 64x64 x 40, 2048 x 64x64 x 128:
    size-independent invariants on every environment (reference environment.py:424-428 uniqueness; obstacles never entered;
    observation channels consistent with the state) + exact oracle comparison on every `stride`-th environment.
    The two larger config-2 launches (16,384 / 32,768 environments) are the ones bench.py's out-of-cache legs time: there the
    AUTOMATIC rules select non-temporal observation stores and four waves per environment (step_nt_store, step_block_threads)."""
    maps, agents, goals, _ = M.generate_scenarios(E, L, N, 0.3, seed=11)
    env = M.VecEnvironment(E, L, N)
    env.load(maps, agents, goals)
    rng = np.random.RandomState(5)
    obs, pos = env.observe()
    tape = np.zeros((T, E, N), np.int8)
    for t in range(T):
        tape[t] = _heuristic_tape_step(_np(obs), rng, 0.8)
        obs, pos, rew, done, rc = env.step(torch.from_numpy(tape[t]).cuda())
    env.check_status()
    p = _np(pos).astype(np.int64)
    assert p.min() >= 0 and p.max() < L
    keys = np.sort(p[..., 0] * L + p[..., 1], axis=1)
    assert np.all(keys[:, 1:] != keys[:, :-1])  # no two agents on a cell
    ee = np.arange(E)[:, None]
    assert maps[ee, p[..., 0], p[..., 1]].sum() == 0  # nobody inside an obstacle
    o = _np(obs)
    assert o.max() <= 1
    assert o[:, :, 0, 4, 4].sum() == 0  # own cell never in the agent channel
    assert o[:, :, 1, 4, 4].sum() == 0  # standing on a free cell
    # agent-channel population == number of other agents within the FOV
    dx = np.abs(p[:, :, None, 0] - p[:, None, :, 0]) <= 4
    dy = np.abs(p[:, :, None, 1] - p[:, None, :, 1]) <= 4
    assert np.array_equal(o[:, :, 0].reshape(E, N, -1).sum(-1), (dx & dy).sum(-1) - 1)
    # rewards are consistent with the classes, and a finished environment pays `finish` to everybody (environment.py:415-419)
    rcl, dn = _np(rc), _np(done).astype(bool)
    assert np.array_equal(_np(rew), np.array([-0.075, 0, -0.075, -0.5, 3], np.float32)[rcl])
    assert np.array_equal((rcl == 4).all(axis=1), dn) and np.array_equal((rcl == 4).any(axis=1), dn)
    assert np.array_equal(dn, (p == goals.astype(np.int64)).all(axis=(1, 2)))
    assert np.array_equal(_np(env.steps()), np.full(E, T, np.int32))
    sample = np.arange(0, E, stride)
    nv = oracle.navi_batch(maps[sample], goals[sample])
    ref = oracle.rollout(maps[sample], agents[sample], goals[sample], nv, tape[:, sample], want_obs_last=True)
    assert np.array_equal(p[sample], ref["final_agents"])
    assert np.array_equal(o[sample], ref["obs_last"])
    assert np.array_equal(rcl[sample], ref["rclass"][-1]) and np.array_equal(dn[sample], ref["done"][-1].astype(bool))
    if E * N * 4 * L * L <= (1 << 30):  # the full navi read-back (one byte per flag) only where it fits comfortably
        assert np.array_equal(_np(env.navi_map())[sample], nv)


@pytest.mark.parametrize("E,L,N", [(64, 32, 40), (256, 20, 6), (64, 10, 1), (16, 64, 128)])
def test_masked_observe_rewrites_only_what_changed(M, E, L, N):
    """mapf_observe_masked (the actor loop's re-observation after an auto-reset): after step + reset_envs(mask) the caller's
    buffers, masked-observed in place, equal a full observe; a workgroup without a flagged environment writes nothing."""
    maps, agents, goals, _ = M.generate_scenarios(E, L, N, 0.2, seed=E + N)
    env = M.VecEnvironment(E, L, N)
    env.load(maps, agents, goals)
    RD = env.obs_bits_row_dwords
    bits = torch.zeros((E, RD), dtype=torch.int32, device="cuda")
    rng = np.random.RandomState(3)
    obs, pos = env.observe(obs_bits_out=bits)
    for t in range(3):
        obs, pos, *_ = env.step(torch.from_numpy(_heuristic_tape_step(_np(obs), rng, 0.8)).cuda(), obs_bits_out=bits)
    mask = torch.from_numpy((rng.random_sample(E) < 0.3).astype(np.uint8)).cuda()
    mask[0], mask[E - 1] = 1, 0
    env.reset_envs(mask, 0.2, seed=99)
    obs, pos = env.observe(obs_bits_out=bits, mask=mask)        # in place, on top of what step() wrote
    got = (_np(obs).copy(), _np(pos).copy(), _np(bits).copy())
    full_obs = torch.empty_like(obs)
    full_bits = torch.empty_like(bits)
    _, full_pos = env.observe(obs_out=full_obs, obs_bits_out=full_bits)
    assert np.array_equal(got[0], _np(full_obs)) and np.array_equal(got[1], _np(full_pos)) and np.array_equal(got[2], _np(full_bits))
    # nothing is written for workgroups without a flagged environment (G environments per workgroup for few agents: <= 8)
    sent = torch.full_like(obs, 7)
    env.observe(obs_out=sent, mask=mask)
    m = _np(mask).astype(bool)
    s = _np(sent)
    assert np.array_equal(s[m], _np(full_obs)[m])
    grp = m.reshape(-1, 8).any(axis=1).repeat(8) if E % 8 == 0 else np.ones(E, bool)
    assert (s[~grp] == 7).all()
    env.check_status()


def test_bad_action_raises_assertion(M):
    maps, agents, goals = H.random_scenarios(2, 8, 4, 0.1, seed=1)
    env = M.VecEnvironment(2, 8, 4)
    env.load(maps, agents, goals)
    acts = torch.zeros((2, 4), dtype=torch.int8, device="cuda")
    acts[1, 2] = 5
    env.step(acts)
    with pytest.raises(AssertionError):
        env.check_status()
    env.step(torch.zeros((2, 4), dtype=torch.int8, device="cuda"))
    env.check_status()  # sticky status was cleared


def test_device_side_load_and_rewind(M):
    maps, agents, goals = H.random_scenarios(8, 16, 12, 0.2, seed=9)
    env = M.VecEnvironment(8, 16, 12)
    env.load(torch.from_numpy(maps).cuda(), torch.from_numpy(agents).cuda(), torch.from_numpy(goals).cuda())
    env.check_status()
    nv = oracle.navi_batch(maps, goals)
    assert np.array_equal(_np(env.navi_map()), nv)
    tape = H.random_tape(10, 8, 12, seed=2)
    outs = []
    for rep in range(2):
        for t in range(10):
            obs, pos, *_ = env.step(torch.from_numpy(tape[t]).cuda())
        outs.append((_np(obs).copy(), _np(pos).copy()))
        env.set_agents(torch.from_numpy(agents).cuda())
        assert int(_np(env.steps()).max()) == 0
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    bad = agents.copy()
    bad[0, 0, 0] = 16
    env.load(torch.from_numpy(maps).cuda(), torch.from_numpy(bad).cuda(), torch.from_numpy(goals).cuda())
    with pytest.raises(Exception):
        env.check_status()


def test_single_env_facade_matches_reference_api(M):
    """`Environment` keeps the reference's signatures / return types (environment.py:198,278,433)."""
    z = H.load_npz("env_fixtures.npz")
    pre = "fix16_c0_"
    m, a, g = z[pre + "map"], z[pre + "agents"].astype(np.int64), z[pre + "goals"].astype(np.int64)
    env = M.Environment()
    env.load(m, a, g)
    assert env.num_agents == 16 and env.map_size == (40, 40) and env.steps == 0
    assert env.navi_map.shape == (16, 4, 48, 48) and env.navi_map.dtype == bool
    obs, pos = env.observe()
    assert obs.shape == (16, 6, 9, 9) and obs.dtype == bool and pos.dtype == np.int64
    acts = z[pre + "mixed_actions"]
    for t in range(8):
        (obs, pos), rewards, done, info = env.step([int(x) for x in acts[t]])
        assert isinstance(rewards, list) and len(rewards) == 16 and isinstance(done, bool)
        assert info == {"step": t}
        assert np.array_equal(pos, z[pre + "mixed_pos"][t + 1])
        assert rewards == z[pre + "mixed_rewards"][t].tolist()
    assert env.steps == 8
    with pytest.raises(AssertionError):
        env.step([0] * 15)
    with pytest.raises(AssertionError):
        env.step([5] + [0] * 15)
    # constructor / reset path: random scenario by the reference's rule
    env2 = M.Environment(map_length=12, num_agents=5)
    o, p = env2.reset()
    assert o.shape == (5, 6, 9, 9) and p.shape == (5, 2)
    o, p = env2.reset(num_agents=3, map_length=10)
    assert o.shape == (3, 6, 9, 9) and env2.map_size == (10, 10)


@pytest.mark.parametrize("E", [512, 40])
def test_multi_handle_launch_equals_per_handle_launches(M, E):
    """mapf_multi_* (include/mapf_env.h): the curriculum's levels -- handles of different (agents, map side) -- stepped, reset and
    re-observed by ONE launch each.  Two identical sets of handles; one is driven through MultiEnvironment, the other handle by
    handle through mapf_step / mapf_observe_masked: every output of every step is identical (byte observations, bit rows,
    positions, reward classes, rewards, done), the trajectories equal the sequential oracle's, and after the merged reset the
    flagged environments hold fresh valid scenarios (navi fields == oracle BFS of the read-back maps / goals) while the others are
    untouched; the merged masked re-observation equals a full observe."""
    from mapf_rl_amd.environment import MultiEnvironment

    levels = [(1, 10), (2, 15), (3, 20), (4, 25), (5, 30), (6, 35), (6, 40), (12, 24), (16, 40), (7, 33)]
    rng = np.random.RandomState(E)
    sets = []
    scen = [M.generate_scenarios(E, L, N, 0.2, seed=100 * N + L)[:3] for N, L in levels]
    for k in range(2):
        envs = []
        for (N, L), (maps, agents, goals) in zip(levels, scen):
            env = M.VecEnvironment(E, L, N)
            env.load(maps, agents, goals)
            envs.append(env)
        sets.append(envs)
    acts = [torch.zeros((E, N), dtype=torch.int8, device="cuda") for N, L in levels]
    bits = [[torch.zeros((E, env.obs_bits_row_dwords), dtype=torch.int32, device="cuda") for env in envs] for envs in sets]
    masks = [torch.zeros(E, dtype=torch.uint8, device="cuda") for _ in levels]
    multi = MultiEnvironment(sets[0], acts, bits[0], masks)
    assert multi.num_workgroups <= sum(E for _ in levels)
    obs = []
    for k in range(2):
        for env, b in zip(sets[k], bits[k]):
            o, _ = env.observe(obs_bits_out=b)
            if k == 0:
                obs.append(o)
    T = 30
    tapes = [np.zeros((T, E, N), np.int8) for N, L in levels]
    for t in range(T):
        for i, (N, L) in enumerate(levels):
            tapes[i][t] = _heuristic_tape_step(_np(obs[i]), rng, (0.5, 0.9, 1.0)[t % 3])
            acts[i].copy_(torch.from_numpy(tapes[i][t]))
        multi.step()
        for i, env in enumerate(sets[1]):
            o, p, r, d, rc = env.step(acts[i], obs_bits_out=bits[1][i])
            a = sets[0][i]
            assert torch.equal(a.obs, o) and torch.equal(a.pos, p) and torch.equal(a.reward, r) and torch.equal(a.done, d), (t, levels[i])
            assert torch.equal(a.reward_class, rc) and torch.equal(bits[0][i], bits[1][i]), (t, levels[i])
    for i, ((N, L), (maps, agents, goals)) in enumerate(zip(levels, scen)):
        sets[0][i].check_status()
        nv = oracle.navi_batch(maps, goals)
        ref = oracle.rollout(maps, agents, goals, nv, tapes[i], want_obs_last=True)
        assert np.array_equal(_np(sets[0][i].agents_pos()), ref["final_agents"]) and np.array_equal(_np(sets[0][i].obs), ref["obs_last"]), levels[i]
        assert np.array_equal(_np(sets[0][i].steps()), np.full(E, T, np.int32))
    # merged reset of the flagged environments, then the merged masked re-observation
    before = [(_np(env.maps()).copy(), _np(env.agents_pos()).copy(), _np(env.goals_pos()).copy()) for env in sets[0]]
    for i, m in enumerate(masks):
        flag = (rng.random_sample(E) < 0.3).astype(np.uint8)
        flag[0], flag[E - 1] = 1, 0
        m.copy_(torch.from_numpy(flag))
    tick = torch.zeros(1, dtype=torch.int64, device="cuda")
    multi.reset(0.2, tick)
    multi.observe_masked()
    for i, ((N, L), env) in enumerate(zip(levels, sets[0])):
        env.check_status()
        flag = _np(masks[i]).astype(bool)
        maps, ag, go = _np(env.maps()), _np(env.agents_pos()), _np(env.goals_pos())
        assert np.array_equal(maps[~flag], before[i][0][~flag]) and np.array_equal(ag[~flag], before[i][1][~flag]) and np.array_equal(go[~flag], before[i][2][~flag])
        assert (np.abs(maps[flag].astype(int) - before[i][0][flag]).sum(axis=(1, 2)) > 0).mean() > 0.9   # fresh maps
        steps = _np(env.steps())
        assert (steps[flag] == 0).all() and (steps[~flag] == T).all()
        ee = np.arange(E)[:, None]
        assert maps[ee, ag[..., 0], ag[..., 1]].sum() == 0 and maps[ee, go[..., 0], go[..., 1]].sum() == 0
        keys = np.sort(np.concatenate([ag[..., 0] * L + ag[..., 1], go[..., 0] * L + go[..., 1]], axis=1).astype(np.int64), axis=1)
        assert np.all(keys[flag][:, 1:] != keys[flag][:, :-1])                 # 2 N distinct cells (environment.py:118-138)
        assert np.array_equal(_np(env.navi_map()), oracle.navi_batch(maps, go))
        full_obs, full_bits = torch.empty_like(env.obs), torch.empty_like(bits[0][i])
        held = env.obs.clone()
        _, full_pos = env.observe(obs_out=full_obs, obs_bits_out=full_bits)
        assert torch.equal(held, full_obs) and torch.equal(bits[0][i], full_bits)
    # two resets draw different scenarios (the set's iteration counter moves the stream)
    m0 = _np(sets[0][0].maps()).copy()
    tick += 1
    multi.reset(0.2, tick)
    multi.observe_masked()
    f0 = _np(masks[0]).astype(bool)
    assert (np.abs(_np(sets[0][0].maps())[f0].astype(int) - m0[f0]).sum(axis=(1, 2)) > 0).mean() > 0.9


def test_multi_handle_set_rejects_wide_shapes(M):
    from mapf_rl_amd import _lib
    from mapf_rl_amd.environment import MultiEnvironment

    env = M.VecEnvironment(8, 32, 40)
    env.reset_envs(None, 0.2, seed=1)
    with pytest.raises(_lib.MapfError) as ex:
        MultiEnvironment([env], [torch.zeros((8, 40), dtype=torch.int8, device="cuda")], [None], [torch.zeros(8, dtype=torch.uint8, device="cuda")])
    assert ex.value.status == _lib.ERR_UNSUPPORTED


def test_two_kernel_reset_draws_the_merged_resets_scenarios(M):
    """mapf_reset_envs builds a scenario in two launches (placement with the map's partitions looked up, then the N navigation fields
    in parallel); the merged launch of a handle set keeps the one-wavefront version that floods from every drawn goal.  Same
    scenario stream, same flags -> the same maps, starts, goals, fields and step counters, bit for bit (also for the environments a
    mask leaves alone), on several shapes up to the merged launch's limit."""
    from mapf_rl_amd.environment import MultiEnvironment

    E = 48
    levels = [(1, 10), (6, 20), (12, 24), (16, 40), (7, 33), (6, 40)]
    seeds = [1000 + 17 * i for i in range(len(levels))]
    rng = np.random.RandomState(5)
    sets = []
    for k in range(2):
        envs = []
        for N, L in levels:
            env = M.VecEnvironment(E, L, N)
            env.load(*M.generate_scenarios(E, L, N, 0.2, seed=100 * N + L)[:3])
            envs.append(env)
        sets.append(envs)
    acts = [torch.zeros((E, N), dtype=torch.int8, device="cuda") for N, L in levels]
    masks = [torch.zeros(E, dtype=torch.uint8, device="cuda") for _ in levels]
    multi = MultiEnvironment(sets[0], acts, [None] * len(levels), masks, reset_seeds=seeds)
    tick = torch.zeros(1, dtype=torch.int64, device="cuda")
    for rnd, density in enumerate((0.2, -1.0, 0.3)):
        for m in masks:
            flag = (rng.random_sample(E) < (1.0 if rnd == 0 else 0.4)).astype(np.uint8)
            m.copy_(torch.from_numpy(flag))
        tick.fill_(3 + rnd)
        multi.reset(density, tick)
        for i, env in enumerate(sets[1]):
            env.reset_envs(masks[i], density, seed=seeds[i] + 3 + rnd)
        for i, (a, b) in enumerate(zip(sets[0], sets[1])):
            a.check_status()
            b.check_status()
            assert np.array_equal(_np(a.maps()), _np(b.maps())), (rnd, levels[i])
            assert np.array_equal(_np(a.agents_pos()), _np(b.agents_pos())) and np.array_equal(_np(a.goals_pos()), _np(b.goals_pos())), (rnd, levels[i])
            assert np.array_equal(_np(a.navi_map()), _np(b.navi_map())) and np.array_equal(_np(a.steps()), _np(b.steps())), (rnd, levels[i])
