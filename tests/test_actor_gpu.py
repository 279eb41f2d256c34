"""GPU: the vectorised actor loop (reference worker.py:368-428) -- episodes recorded on the device match an
independent re-simulation (CPU oracle environment + the same network in fp32 is too loose for bf16 argmax, so
the check is structural): recorded observations are exactly the env's observations along the recorded
actions, priorities follow LocalBuffer.finish, and the replay ring receives what was recorded."""
import numpy as np
import pytest
import torch

from oracle import oracle, replay_oracle as RO
from tests import helpers as H

pytestmark = pytest.mark.gpu


def _unpack_rows(bits_i32, N):
    b = bits_i32.cpu().numpy().view(np.uint32)
    R = b.shape[0]
    raw = np.unpackbits(b.view(np.uint8).reshape(R, -1), axis=1, bitorder="little")[:, :N * 486]
    return raw.reshape(R, N, 6, 9, 9)


def test_actor_records_consistent_episodes():
    import mapf_rl_amd as M
    from mapf_rl_amd.actor import VecActor
    from mapf_rl_amd.model import Network
    from mapf_rl_amd.replay import GlobalBuffer

    torch.manual_seed(0)
    E, L, N = 12, 10, 3
    maps, agents, goals, _ = M.generate_scenarios(E, L, N, 0.2, seed=5)
    env = M.VecEnvironment(E, L, N)
    env.load(maps, agents, goals)
    net = Network().cuda().eval()
    buf = GlobalBuffer(32, max_agents=N)
    actor = VecActor(env, net, buf, epsilons=0.3, max_steps=24, seed=3, density=0.2, keep_flushed=True)
    # remember the scenario each env is playing so finished episodes can be re-simulated
    scen = {e: (maps[e].copy(), agents[e].copy(), goals[e].copy()) for e in range(E)}
    checked = 0
    for it in range(80):
        n_before = len(actor.flushed)
        actor.step()
        for ep in actor.flushed[n_before:]:
            e, size = ep["env"], ep["size"]
            m, a, g = scen[e]
            nv = oracle.navi(m, g)
            obs = _unpack_rows(ep["obs"], N)
            assert np.array_equal(obs[0], oracle.observe(m, a, nv))
            # replaying agent 0's recorded action is not enough (other agents' actions are not stored: Q7),
            # so check what IS determined: row 0, sizes, flags, priorities, rewards in the reward set
            assert 1 <= size <= 24 and (ep["done"] or size == 24)
            rew = ep["rew"].cpu().numpy()
            assert set(np.unique(rew.astype(np.float32))) <= set(np.array([-0.075, -0.5, 0.0, 3.0], np.float16).astype(np.float32))
            td = RO.local_finish(ep["q"].cpu().numpy(), ep["act"].cpu().numpy(), rew, size, capacity=24)
            assert ep["td"].shape == (256,) and float(ep["td"][24:].abs().sum()) == 0.0
            assert np.allclose(ep["td"].cpu().numpy()[:24], td, rtol=1e-12, atol=1e-12)
            if ep["done"]:
                assert np.all(rew[-1:] == np.float16(3.0)) and int(ep["comm"][size].abs().sum()) == 0
            else:
                assert torch.equal(ep["comm"][size], ep["comm"][size - 1])
            checked += 1
            # the env restarted on a new scenario: read it back
            scen[e] = (env.maps()[e].cpu().numpy(), env.agents_pos()[e].cpu().numpy() if False else None, env.goals_pos()[e].cpu().numpy())
            scen[e] = (scen[e][0], _start_of(actor, e, N, L), scen[e][2])
    assert checked >= E  # every env finished at least one episode (max_steps = 24, 80 iterations)
    assert len(buf) == sum(ep["size"] for ep in actor.flushed[-min(len(actor.flushed), 32):]) or len(actor.flushed) > 32
    assert actor.env_steps == 80 * E
    # the ring is sampleable and the learner's window gather works on actor-produced data: every drawn leaf
    # lies inside its episode (reference assert, worker.py:120)
    for _ in range(20):
        out = buf.sample_batch(64)
        assert out[0].shape == (64, 18, N, 6, 9, 9) and torch.isfinite(out[9]).all()
        assert float(out[4].min()) >= 1 and float(out[4].max()) <= 2


def _start_of(actor, e, N, L):
    """start positions of env e's current episode = positions of agents in its first observation row: recover from
    the env's current state only when t == 0, which holds right after the reset."""
    assert int(actor.t[e]) == 0
    return actor.env.agents_pos()[e].cpu().numpy()


def test_full_trajectory_resimulation_single_agent_greedy():
    """N = 1, epsilon = 0: the only agent's action is stored, so whole episodes can be re-simulated by the CPU
    oracle from the recorded actions and must reproduce every recorded observation row and reward."""
    import mapf_rl_amd as M
    from mapf_rl_amd.actor import VecActor
    from mapf_rl_amd.model import Network

    torch.manual_seed(1)
    E, L, N = 16, 8, 1
    maps, agents, goals, _ = M.generate_scenarios(E, L, N, 0.15, seed=9)
    env = M.VecEnvironment(E, L, N)
    env.load(maps, agents, goals)
    actor = VecActor(env, Network().cuda().eval(), None, epsilons=0.5, max_steps=12, seed=1, density=0.15, keep_flushed=True)
    for _ in range(12):
        actor.step()
    first = {}
    for ep in actor.flushed:
        first.setdefault(ep["env"], ep)
    assert len(first) == E  # 12 iterations with max_steps 12: every env flushed its first episode
    for e, ep in first.items():
        m, a, g = maps[e], agents[e].copy(), goals[e]
        nv = oracle.navi(m, g)
        obs = _unpack_rows(ep["obs"], N)
        ag = a.astype(np.int16)
        assert np.array_equal(obs[0], oracle.observe(m, ag, nv))
        for t in range(ep["size"]):
            st, ag, rc, dn = oracle.step(m, ag, g, ep["act"].cpu().numpy()[t:t + 1].astype(np.int8))
            assert st == 0
            assert np.array_equal(obs[t + 1], oracle.observe(m, ag, nv)), (e, t)
            assert np.float16(H.REWARD_VALUES[rc[0]]) == ep["rew"].cpu().numpy()[t]
            assert dn == (ep["done"] and t == ep["size"] - 1)


def _det_net():
    from mapf_rl_amd.model import Network

    net = Network().cuda().eval()
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    net.load_state_dict({k: torch.from_numpy(v) for k, v in H.det_state_dict(shapes, seed=1234).items()})
    return net


@pytest.mark.parametrize("tag", ["to", "dn"])
def test_actor_tape_matches_reference(tag):
    """The reference's actor loop (worker.py:376-407, epsilon = 0, deterministic weights) recorded by its own LocalBuffer
    (buffer.py:108-179): tests/golden/dqn_actor.npz, episode 'to' (policy actions, ends at the step limit: quirk Q8, the last comm
    row comes from a model.step on the stale observation) and 'dn' (scripted actions, ends with `done`).  VecActor replays the
    reference's joint actions (teacher forcing) through the fused bf16 kernels and must record, for agent 0 (quirk Q7):
    observations, actions, rewards, comm rows, size, done EXACTLY; hidden states, Q-values and the initial priorities (td) within
    the bf16 tolerance; and its own greedy action must be the reference's wherever the reference's top-2 Q gap exceeds it."""
    import mapf_rl_amd as M
    from mapf_rl_amd.actor import VecActor
    from mapf_rl_amd.replay import GlobalBuffer

    z = H.load_npz("dqn_actor.npz")
    pre = tag + "_"
    m, a, g = z[pre + "map"], z[pre + "agents"], z[pre + "goals"]
    L, N, size, ms = m.shape[0], a.shape[0], int(z[pre + "size"]), int(z[pre + "max_steps"])
    env = M.VecEnvironment(1, L, N)
    env.load(m[None], a[None], g[None])
    buf = GlobalBuffer(4, max_agents=N)
    actor = VecActor(env, _det_net(), buf, epsilons=0.0, max_steps=ms, seed=0, density=0.2, keep_flushed=True)
    acts = z[pre + "actions"]
    TOL = 2e-2
    for t in range(size):
        assert not actor.flushed
        assert np.array_equal(actor.pos[0].cpu().numpy(), z[pre + "pos"][t])
        actor.step(actions_override=torch.from_numpy(acts[t:t + 1].astype(np.int64)))
        clear = z[pre + "gap"][t] > 2 * TOL
        if tag == "to":  # the reference executed its own greedy actions here
            assert np.array_equal(actor.last_policy_actions[0].cpu().numpy()[clear], acts[t][clear]), t
    assert len(actor.flushed) == 1
    ep = actor.flushed[0]
    assert ep["size"] == size and ep["done"] == bool(z[pre + "done"])
    ref_obs = H.unpack_bits(z[pre + "obs_bits"], (size + 1, N, 6, 9, 9))
    assert np.array_equal(_unpack_rows(ep["obs"], N), ref_obs)                                   # every observation row
    assert np.array_equal(ep["act"].cpu().numpy(), z[pre + "act_buf"])
    assert np.array_equal(ep["rew"].cpu().numpy().astype(np.float32), z[pre + "rew_buf"])         # f16 values, exact
    comm = ep["comm"].cpu().numpy().view(np.uint32)                                              # [size + 1, A, CW] bit rows
    comm_bool = np.unpackbits(comm.view(np.uint8).reshape(size + 1, N, -1), axis=2, bitorder="little")[:, :, :N].astype(bool)
    assert np.array_equal(comm_bool, z[pre + "comm_buf"])      # incl. the last row: zeros after `done`, the stale-observation mask on time-out
    close = lambda x, y, tol: np.all(np.abs(np.asarray(x, np.float64) - y) <= tol * np.maximum(1.0, np.abs(y)))
    assert close(ep["q"].cpu().numpy(), z[pre + "q0"], TOL)
    assert close(ep["hid"].float().cpu().numpy(), z[pre + "hid_buf0"], TOL)
    td = ep["td"].cpu().numpy()
    assert td.shape == (256,) and not td[size:].any()
    assert close(td[:size], z[pre + "td"][:size], 2 * TOL)     # |r + 0.99 r' + max Q - Q(a)|: two Q-values
    # and the reference's own numbers through the same formula reproduce its td (the recording logic, fp-exact)
    ref_td = RO.local_finish(np.concatenate([z[pre + "q0"], np.zeros((1, 5), np.float32)]), z[pre + "act_buf"], z[pre + "rew_buf"].astype(np.float16), size, capacity=ms)
    assert np.allclose(ref_td[:size], z[pre + "td"][:size], rtol=0, atol=1e-12)


def test_latent_cache_encodes_only_changed_rows_and_is_exact():
    """fused.LatentCache (mapf_obs_changed + mapf_encoder_forward_rows): with some agents' observations unchanged from call to call,
    the cached latents are the same bits as encoding every row; the changed rows are counted on the device; a weight change or a
    different observation buffer invalidates the cache."""
    from mapf_rl_amd.fused import LatentCache, PackedEncoder, encoder_forward
    from mapf_rl_amd.model import Network

    torch.manual_seed(0)
    net = Network().cuda()
    packed, cache = PackedEncoder(), LatentCache()
    g = torch.Generator(device="cuda").manual_seed(1)
    R = 1003  # odd: rows at every 2-byte alignment, a ragged last workgroup
    obs = (torch.rand((R, 6, 9, 9), device="cuda", generator=g) < 0.3).to(torch.uint8)
    for k in range(6):
        if k:
            change = torch.rand(R, device="cuda", generator=g) < (0.0, 0.3, 1.0, 0.05, 0.0, 0.5)[k]
            new = (torch.rand((R, 6, 9, 9), device="cuda", generator=g) < 0.3).to(torch.uint8)
            new[:, 0, 0, 0] = 1 - obs[:, 0, 0, 0]  # a changed row really differs
            obs.copy_(torch.where(change.view(R, 1, 1, 1), new, obs))
        lat = cache.encode(obs, packed, net.obs_encoder)
        wp, bp = packed.get(net.obs_encoder)
        assert torch.equal(lat, encoder_forward(obs, wp, bp)), k
        assert cache.last_encoded() == (R if k == 0 else int(change.sum())), k
    assert cache.full == 1 and cache.calls == 6
    with torch.no_grad():
        net.obs_encoder[0].bias.add_(0.1)  # new weights: everything is encoded again
    lat = cache.encode(obs, packed, net.obs_encoder)
    wp, bp = packed.get(net.obs_encoder)
    assert cache.full == 2 and cache.last_encoded() == R and torch.equal(lat, encoder_forward(obs, wp, bp))
    other = obs.clone()
    cache.encode(other, packed, net.obs_encoder)
    assert cache.full == 3


@pytest.mark.parametrize("R", [1, 2, 3, 15, 16, 17, 33, 4099])
def test_obs_changed_lists_exactly_the_changed_rows(R):
    """mapf_obs_changed at row counts around its work units (2 rows per wavefront step, 16 per list-slot atomic): the list holds
    exactly the changed rows (any order), the packed buffer their bytes in list order at the 488-byte stride, `prev` is refreshed."""
    from mapf_rl_amd._lib import check, lib
    from mapf_rl_amd.fused import ENC_PACKED_OBS_STRIDE

    g = torch.Generator(device="cuda").manual_seed(R)
    prev = (torch.rand((R, 486), device="cuda", generator=g) < 0.3).to(torch.uint8)
    for frac in (0.0, 0.4, 1.0):
        obs = prev.clone()
        change = torch.rand(R, device="cuda", generator=g) < frac
        col = torch.randint(0, 486, (R,), device="cuda", generator=g)  # one differing byte anywhere in the row is enough
        rows = change.nonzero().view(-1)
        obs[rows, col[rows]] ^= 1
        lst = torch.full((R,), -1, dtype=torch.int32, device="cuda")
        cnt = torch.full((1,), -1, dtype=torch.int32, device="cuda")
        packed = torch.full((R, ENC_PACKED_OBS_STRIDE), 255, dtype=torch.uint8, device="cuda")
        check(lib.mapf_obs_changed(obs.data_ptr(), prev.data_ptr(), R, lst.data_ptr(), cnt.data_ptr(), packed.data_ptr(), None), "mapf_obs_changed")
        n = int(cnt)
        assert n == int(change.sum())
        got = lst[:n].long()
        assert torch.equal(got.sort().values, rows) and bool((lst[n:] == -1).all())
        assert torch.equal(packed[:n, :486], obs[got]) and bool((packed[n:] == 255).all())
        assert torch.equal(prev, obs)


def test_fused_iteration_tail_equals_the_separate_calls():
    """VecActor.FUSED_TAIL (mapf_actor_iteration_tail: exploration .. episode flush as one library call) against the eight separate
    calls from the same seeds, with exploration on: same executed and greedy actions, hidden states, local buffers, replay ring and
    sum tree, episode counters, environment state -- bit for bit."""
    import mapf_rl_amd as M
    from mapf_rl_amd.actor import VecActor
    from mapf_rl_amd.model import Network
    from mapf_rl_amd.replay import GlobalBuffer

    E, L, N, K = 96, 14, 5, 45
    out = {}
    try:
        for fused in (True, False):
            VecActor.FUSED_TAIL = fused
            torch.manual_seed(4)
            net = Network().cuda()
            env = M.VecEnvironment(E, L, N)
            maps, agents, goals, _ = M.generate_scenarios(E, L, N, 0.25, seed=9)
            env.load(maps, agents, goals)
            buf = GlobalBuffer(256, max_agents=N + 1, init_set=(N, L), fixed_level=True)  # (rows wider than the level: the padded layout)
            actor = VecActor(env, net, buf, epsilons=0.4, seed=6, density=0.25, max_steps=10)
            acts, fins = [], []
            for _ in range(K):
                fins.append(actor.step().clone())
                acts.append(actor.last_policy_actions.clone())
            torch.cuda.synchronize()
            out[fused] = dict(acts=torch.stack(acts), fins=torch.stack(fins), hidden=actor.hidden.clone(), tree=buf.priority_tree.tree().clone(),
                              state=buf.state(), t=actor.t.clone(), lb_obs=actor.lb_obs.clone(), lb_act=actor.lb_act.clone(), lb_q=actor.lb_q.clone(),
                              lb_comm=actor.lb_comm.clone(), obs=actor.obs.clone(), pos=actor.pos.clone(), counters=actor.counters.clone(),
                              log=actor.stat_log.clone(), steps=actor.env_steps)
    finally:
        VecActor.FUSED_TAIL = True
    a, b = out[True], out[False]
    assert a["state"] == b["state"] and a["steps"] == b["steps"] and int(a["counters"][0]) > E  # several episodes per environment
    for k in a:
        if torch.is_tensor(a[k]):
            assert torch.equal(a[k], b[k]), k


def test_explore_kernel_statistics_and_determinism():
    """mapf_actor_explore (worker.py:380-382): agent 0 of environment e takes a uniform action with probability eps[e], everybody
    else keeps the greedy action; the int8 copy equals the result, the greedy copy the input; draws depend on (seed, counter) only."""
    from mapf_rl_amd._lib import check, lib

    E, N = 40000, 3
    g = torch.Generator(device="cuda").manual_seed(0)
    greedy = torch.randint(0, 5, (E, N), device="cuda", generator=g)
    eps = torch.full((E,), 0.25, dtype=torch.float64, device="cuda")
    eps[: E // 2] = 0.0

    def run(seed, counter):
        a, pol, a8 = greedy.clone(), torch.full_like(greedy, -1), torch.full((E, N), -1, dtype=torch.int8, device="cuda")
        check(lib.mapf_actor_explore(E, N, a.data_ptr(), pol.data_ptr(), a8.data_ptr(), eps.data_ptr(), seed, counter, None), "mapf_actor_explore")
        assert torch.equal(pol, greedy) and torch.equal(a8.long(), a) and torch.equal(a[:, 1:], greedy[:, 1:])
        assert int(a.min()) >= 0 and int(a.max()) <= 4
        return a[:, 0]

    a = run(7, 0)
    assert torch.equal(a[: E // 2], greedy[: E // 2, 0])  # eps = 0: never
    changed = float((a[E // 2:] != greedy[E // 2:, 0]).float().mean())  # a uniform draw equals the greedy action 1 time in 5
    assert abs(changed - 0.25 * 0.8) < 0.015, changed
    drawn = a[E // 2:][a[E // 2:] != greedy[E // 2:, 0]]
    assert all(abs(float((drawn == k).float().mean()) - 0.2) < 0.03 for k in range(5))
    assert torch.equal(a, run(7, 0)) and not torch.equal(a, run(7, 1)) and not torch.equal(a, run(8, 0))


def test_actor_with_latent_reuse_records_the_same_episodes():
    """VecActor.REUSE_LATENTS on / off from the same seeds: identical actions, Q-values, hidden states and replay contents (the
    encoder is per observation; an unchanged observation has an unchanged latent).  Also the reference's actor semantics
    (worker.py:416-420): with weights_period the actor acts on its own snapshot until the next pull."""
    import mapf_rl_amd as M
    from mapf_rl_amd.actor import VecActor
    from mapf_rl_amd.model import Network
    from mapf_rl_amd.replay import GlobalBuffer

    E, L, N, K = 64, 16, 10, 40
    out = {}
    try:
        for reuse in (True, False):
            VecActor.REUSE_LATENTS = reuse
            torch.manual_seed(3)
            net = Network().cuda()
            env = M.VecEnvironment(E, L, N)
            maps, agents, goals, _ = M.generate_scenarios(E, L, N, 0.2, seed=8)
            env.load(maps, agents, goals)
            buf = GlobalBuffer(256, max_agents=N, init_set=(N, L), fixed_level=True)
            actor = VecActor(env, net, buf, seed=5, density=0.2, max_steps=16)
            assert (actor.latents is not None) == reuse
            acts, enc = [], []
            for _ in range(K):
                actor.step()
                acts.append(actor.last_policy_actions.clone())
                if reuse:
                    enc.append(actor.latents.last_encoded())
            torch.cuda.synchronize()
            out[reuse] = (torch.stack(acts), actor.hidden.clone(), buf.priority_tree.tree().clone(), buf.state(), enc)
    finally:
        VecActor.REUSE_LATENTS = True
    a, b = out[True], out[False]
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]) and a[3] == b[3]
    assert a[4][0] == E * N and min(a[4][1:]) < E * N  # standing agents are not encoded again
    # weight snapshots: the actor keeps acting on the old weights until the pull
    net = Network().cuda()
    env = M.VecEnvironment(8, 12, 4)
    env.reset_envs(None, 0.2, seed=1)
    actor = VecActor(env, net, None, seed=0, density=0.2, weights_period=3)
    assert actor.model is not net
    with torch.no_grad():
        net.adv.bias.add_(5.0)
    before = actor.model.adv.bias.detach().clone()
    for k in range(1, 8):
        actor.step()
        pulled = k > 3
        assert torch.equal(actor.model.adv.bias, net.adv.bias if pulled else before), k


def test_actor_with_staged_scenarios_equals_the_actor_that_draws_at_the_reset():
    """VecActor.STAGE_AHEAD: the next scenarios are drawn on a second stream beside the policy's forward pass and an episode end only
    swaps them in.  The same actor with the staging stream taken away draws them at the reset: same episodes, same replay, bit for bit."""
    import mapf_rl_amd as M
    from mapf_rl_amd.actor import VecActor
    from mapf_rl_amd.model import Network
    from mapf_rl_amd.replay import GlobalBuffer

    def run(staged):
        torch.manual_seed(3)
        E, L, N = 48, 12, 4
        env = M.VecEnvironment(E, L, N)
        maps, agents, goals, _ = M.generate_scenarios(E, L, N, 0.2, seed=5)
        env.load(maps, agents, goals)
        buf = GlobalBuffer(128, max_agents=6, init_set=(N, L), fixed_level=True)
        actor = VecActor(env, Network().cuda(), buf, seed=1, density=0.2, max_steps=24)
        if not staged:
            actor._stage_stream = None
        for _ in range(120):
            actor.step()
        torch.cuda.synchronize()
        env.check_status()
        return env.maps().clone(), env.agents_pos().clone(), env.goals_pos().clone(), actor.obs.clone(), buf.state(), buf.priority_tree.tree().clone(), actor.episodes

    a, b = run(True), run(False)
    assert a[6] > 100 and a[4] == b[4] and a[6] == b[6]
    for x, y in zip(a[:4], b[:4]):
        assert torch.equal(x, y)
    assert torch.equal(a[5], b[5])
