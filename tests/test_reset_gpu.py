"""GPU: on-device scenario generation / auto-reset (mapf_reset_envs; reference Environment.reset,
environment.py:146-196).  The RNG differs from the reference's, so parity is structural + statistical:
every invariant the reference guarantees by construction, the navi fields bit-exact against the oracle for the
generated scenario, and placement statistics against the host generator that implements the same rule."""
import numpy as np
import pytest
import torch

from oracle import oracle
from tests import helpers as H

pytestmark = pytest.mark.gpu


def _np(t):
    return t.cpu().numpy()


def _check_invariants(maps, agents, goals, L, N):
    E = maps.shape[0]
    assert set(np.unique(maps)) <= {0, 1}
    for e in range(E):
        cells = np.concatenate([agents[e], goals[e]]).astype(np.int64)
        assert cells.min() >= 0 and cells.max() < L
        assert len(np.unique(cells[:, 0] * L + cells[:, 1])) == 2 * N          # all 2N cells distinct
        assert maps[e][cells[:, 0], cells[:, 1]].sum() == 0                     # and free


@pytest.mark.parametrize("E,L,N,rho", [(256, 32, 40, 0.3), (64, 40, 16, -1.0), (64, 10, 1, -1.0), (32, 64, 40, 0.3), (64, 16, 40, 0.3)])
def test_reset_all_invariants_and_navi(E, L, N, rho):
    import mapf_rl_amd as M

    env = M.VecEnvironment(E, L, N)
    env.reset_envs(None, rho, seed=7)
    env.check_status()
    maps, agents, goals = _np(env.maps()), _np(env.agents_pos()), _np(env.goals_pos())
    _check_invariants(maps, agents, goals, L, N)
    nv = oracle.navi_batch(maps, goals)
    assert np.array_equal(_np(env.navi_map()), nv)                               # fused BFS == oracle BFS
    for e in range(min(E, 16)):                                                  # goal reachable from start
        for i in range(min(N, 6)):
            d = oracle.dist(maps[e], goals[e, i])
            assert d[agents[e, i, 0], agents[e, i, 1]] < 2147483647
    assert int(_np(env.steps()).max()) == 0
    # the generated state is steppable and matches the oracle
    tape = H.random_tape(8, E, N, seed=3)
    for t in range(8):
        obs, pos, *_ = env.step(torch.from_numpy(tape[t]).cuda())
    env.check_status()
    ref = oracle.rollout(maps, agents, goals, nv, tape, want_obs_last=True)
    assert np.array_equal(_np(pos), ref["final_agents"]) and np.array_equal(_np(obs), ref["obs_last"])
    if rho > 0:
        assert abs(maps.mean() - rho) < 0.02


def test_masked_reset_touches_only_flagged_envs():
    import mapf_rl_amd as M

    E, L, N = 32, 20, 6
    env = M.VecEnvironment(E, L, N)
    env.reset_envs(None, 0.25, seed=1)
    tape = H.random_tape(3, E, N, seed=1)
    for t in range(3):
        env.step(torch.from_numpy(tape[t]).cuda())
    m0, a0, g0, s0, n0 = _np(env.maps()), _np(env.agents_pos()), _np(env.goals_pos()), _np(env.steps()), _np(env.navi_map())
    mask = torch.zeros(E, dtype=torch.uint8)
    mask[[3, 4, 17]] = 1
    env.reset_envs(mask.cuda(), 0.25, seed=2)
    env.check_status()
    m1, a1, g1, s1, n1 = _np(env.maps()), _np(env.agents_pos()), _np(env.goals_pos()), _np(env.steps()), _np(env.navi_map())
    keep = np.ones(E, bool)
    keep[[3, 4, 17]] = False
    assert np.array_equal(m0[keep], m1[keep]) and np.array_equal(a0[keep], a1[keep]) and np.array_equal(g0[keep], g1[keep])
    assert np.array_equal(n0[keep], n1[keep]) and np.array_equal(s1[keep], np.full(keep.sum(), 3)) and np.all(s1[~keep] == 0)
    assert not np.array_equal(m0[~keep], m1[~keep])
    # a second reset of the same env with the same seed draws a NEW scenario (per-env reset counter)
    env.reset_envs(mask.cuda(), 0.25, seed=2)
    assert not np.array_equal(_np(env.maps())[~keep], m1[~keep])


def test_statistics_match_host_generator():
    """Same rule as mapf_generate (host): map density, triangular density, start-goal distance distribution."""
    import mapf_rl_amd as M

    E, L, N = 512, 24, 8
    env = M.VecEnvironment(E, L, N)
    env.reset_envs(None, -1.0, seed=11)
    maps, agents, goals = _np(env.maps()), _np(env.agents_pos()), _np(env.goals_pos())
    hm, ha, hg, _ = M.generate_scenarios(E, L, N, -1.0, seed=12)
    assert abs(maps.mean() - hm.mean()) < 0.02 and abs(maps.mean() - (0.33 + 0.5) / 3) < 0.03

    def mean_dist(mm, aa, gg):
        tot, cnt = 0, 0
        for e in range(0, E, 4):
            for i in range(N):
                d = oracle.dist(mm[e], gg[e, i])[aa[e, i, 0], aa[e, i, 1]]
                tot += d
                cnt += 1
        return tot / cnt

    d_dev, d_host = mean_dist(maps, agents, goals), mean_dist(hm, ha, hg)
    assert abs(d_dev - d_host) < 0.12 * d_host, (d_dev, d_host)


def test_infeasible_shape_sets_status():
    import mapf_rl_amd as M

    env = M.VecEnvironment(2, 4, 40)   # 16 cells cannot host 80 distinct positions
    env.reset_envs(None, 0.3, seed=0)
    with pytest.raises(Exception):
        env.check_status()


@pytest.mark.parametrize("N,L,E", [(6, 20, 6000), (40, 32, 4000), (16, 40, 4000)])
def test_reset_kernel_matches_reference_statistics(N, L, E):
    """reset_kernel against statistics of the REFERENCE's generator (tests/golden/gen_stats.npz; reference
    environment.py:21-70,100-138): realised density under triangular(0, 0.33, 0.5), component structure, start-goal BFS /
    Manhattan distance, share of agents in the largest component.  Two-sample KS <= 0.05 (tests/gen_stats.py)."""
    import mapf_rl_amd as M
    from tests import gen_stats as GS

    env = M.VecEnvironment(E, L, N)
    env.reset_envs(None, -1.0, seed=23)
    env.check_status()
    GS.check(N, L, GS.batch_stats(_np(env.maps()), _np(env.agents_pos()), _np(env.goals_pos())))


@pytest.mark.parametrize("E,L,N,rho", [(96, 32, 40, 0.3), (64, 40, 16, -1.0), (48, 64, 40, 0.3), (128, 10, 1, -1.0)])
def test_staged_scenarios_are_the_direct_resets_scenarios(E, L, N, rho):
    """Round 5 (mapf_stage_next): the scenario of (seed, environment, reset count) is drawn AHEAD -- on another stream, beside whatever
    the caller runs -- and a reset only hands it over.  Two handles go through the same sequence of masked resets, one staging
    ahead, one drawing at the reset: maps, starts, goals, navigation fields and step counters agree bit for bit after every round,
    including environments reset in consecutive rounds, never reset, and a round with nothing flagged."""
    import mapf_rl_amd as M

    a, b = M.VecEnvironment(E, L, N), M.VecEnvironment(E, L, N)
    side = torch.cuda.Stream()
    cur = torch.cuda.current_stream()
    rng = np.random.RandomState(E + L)
    a.reset_envs(None, rho, seed=11)
    b.reset_envs(None, rho, seed=11)
    for rnd in range(7):
        mask = torch.from_numpy((rng.random_sample(E) < (0.0 if rnd == 3 else 0.35)).astype(np.uint8)).cuda()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            a.stage_next(rho, seed=11)
            ev = torch.cuda.Event()
            ev.record(side)
        cur.wait_event(ev)
        a.reset_envs(mask, rho, seed=11)      # hands the staged scenarios over
        b.reset_envs(mask, rho, seed=11)      # draws them now
        a.check_status()
        b.check_status()
        for name in ("maps", "agents_pos", "goals_pos", "navi_map"):
            assert torch.equal(getattr(a, name)(), getattr(b, name)()), (rnd, name)
    _check_invariants(_np(a.maps()), _np(a.agents_pos()), _np(a.goals_pos()), L, N)


def test_reset_without_a_staged_scenario_is_reported():
    """Two resets of one environment with a single staging call in between: the second finds nothing staged for its epoch -> sticky
    MAPF_ERR_NOT_READY (the caller's ordering contract, include/mapf_env.h); another (density, seed) takes the direct path."""
    import mapf_rl_amd as M
    from mapf_rl_amd._lib import ERR_NOT_READY, MapfError

    env = M.VecEnvironment(8, 16, 4)
    env.reset_envs(None, 0.2, seed=3)
    env.stage_next(0.2, seed=3)
    mask = torch.ones(8, dtype=torch.uint8, device="cuda")
    env.reset_envs(mask, 0.2, seed=3)
    env.check_status()
    env.reset_envs(mask, 0.2, seed=3)
    with pytest.raises(MapfError) as ex:
        env.check_status()
    assert ex.value.status == ERR_NOT_READY
    env.reset_envs(mask, 0.2, seed=4)        # not the staged stream: drawn directly
    env.check_status()


def test_staged_resets_at_the_full_baseline_size():
    """BASELINE config 2's launch (4096 environments, 32x32, 40 agents): three rounds of staged resets with a third of the
    environments flagged each time -- the invariants the reference guarantees by construction for every environment, navigation
    fields equal to the oracle's on a sample, step counters zeroed exactly where flagged, and no environment left without a staged
    scenario (sticky status clean)."""
    import mapf_rl_amd as M

    E, L, N = 4096, 32, 40
    env = M.VecEnvironment(E, L, N)
    env.reset_envs(None, 0.3, seed=21)
    rng = np.random.RandomState(9)
    acts = torch.zeros((E, N), dtype=torch.int8, device="cuda")
    for rnd in range(3):
        env.stage_next(0.3, seed=21)
        env.step(acts)                                   # every environment's step counter moves to 1
        mask = torch.from_numpy((rng.random_sample(E) < 0.33).astype(np.uint8)).cuda()
        before = _np(env.maps())
        env.reset_envs(mask, 0.3, seed=21)
        env.check_status()
        steps = _np(env.steps())
        after = _np(env.maps())
        m = _np(mask).astype(bool)
        assert np.all(steps[m] == 0) and np.all(steps[~m] >= 1)   # step counters zeroed exactly where flagged
        assert np.array_equal(after[~m], before[~m])     # unflagged environments untouched
        assert (after[m] != before[m]).reshape(m.sum(), -1).any(axis=1).mean() > 0.99   # flagged ones re-drawn
    maps, agents, goals = _np(env.maps()), _np(env.agents_pos()), _np(env.goals_pos())
    _check_invariants(maps[:512], agents[:512], goals[:512], L, N)
    sample = np.arange(0, E, 64)
    assert np.array_equal(_np(env.navi_map())[sample], oracle.navi_batch(maps[sample], goals[sample]))
