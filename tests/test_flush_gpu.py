"""GPU: the batched, device-side episode flush (include/mapf_replay.h mapf_replay_add_many through VecActor._flush):
every environment that finished in a step is appended to the replay ring in one launch sequence with no host read.
Parity: the reference's GlobalBuffer.add (worker.py:71-104) fed the same episodes ONE BY ONE in environment order
(oracle/replay_oracle.py, golden-pinned in tests/test_replay_oracle.py) -- ring pointer, size, counter, the whole sum tree,
and sampled windows -- including a 4096-environment simultaneous time-out with fewer ring slots than episodes."""
import numpy as np
import pytest
import torch

from oracle import replay_oracle as RO

pytestmark = pytest.mark.gpu


def _unpack_rows(bits_i32, N):
    b = bits_i32.cpu().numpy().view(np.uint32)
    R = b.shape[0]
    raw = np.unpackbits(b.view(np.uint8).reshape(R, -1), axis=1, bitorder="little")[:, :N * 486]
    return raw.reshape(R, N, 6, 9, 9).astype(bool)


def _unpack_comm(bits_i32, N):
    b = bits_i32.cpu().numpy().view(np.uint32)  # [R, A, CW]
    R, A = b.shape[:2]
    return np.unpackbits(b.view(np.uint8).reshape(R, A, -1), axis=2, bitorder="little")[:, :N, :N].astype(bool)


@pytest.mark.parametrize("E,cap,steps", [(4096, 512, 8), (1024, 1024, 8), (96, 64, 21)])
def test_simultaneous_flush_equals_sequential_reference_adds(E, cap, steps):
    import mapf_rl_amd as M
    from mapf_rl_amd.actor import VecActor
    from mapf_rl_amd.model import Network
    from mapf_rl_amd.replay import GlobalBuffer

    torch.manual_seed(E)
    L, N, S = 10, 2, 8
    maps, agents, goals, _ = M.generate_scenarios(E, L, N, 0.1, seed=3)
    env = M.VecEnvironment(E, L, N)
    env.load(maps, agents, goals)
    buf = GlobalBuffer(cap, max_agents=N)
    actor = VecActor(env, Network().cuda().eval(), buf, epsilons=0.6, max_steps=S, seed=5, density=0.1, keep_flushed=True)
    ref = RO.GlobalBuffer(cap, max_agents=N)
    fed = 0
    for it in range(steps):
        fin = actor.step()
        if it == S - 1:
            assert int(fin.sum()) > 0.9 * E  # the simultaneous time-out (some environments finished earlier and restarted)
        for ep in actor.flushed[fed:]:     # episodes in device order: step by step, ascending environment index
            size = ep["size"]
            hid = np.repeat(ep["hid"].cpu().numpy()[:, None], N, axis=1)
            ref.add(N, _unpack_rows(ep["obs"], N), ep["act"].cpu().numpy(), ep["rew"].cpu().numpy(), hid, ep["td"].cpu().numpy(),
                    ep["done"], size, _unpack_comm(ep["comm"], N), zero_padding=True)
        fed = len(actor.flushed)
    assert fed >= E and actor.episodes == fed
    ptr, size, counter, _ = buf.state()
    assert (ptr, size) == (ref.ptr, ref.size) and counter == sum(ep["size"] for ep in actor.flushed)
    tree = buf.priority_tree.tree().cpu().numpy()
    # leaves: td ** 0.6 from the kernel's own f64 td (LocalBuffer.finish) vs numpy on the torch-formula td; ancestors are sums
    assert np.allclose(tree, ref.tree.tree, rtol=1e-13, atol=0)
    rng = np.random.RandomState(1)
    B = 64
    u = rng.random_sample(B) * (ref.tree.tree[0] / B)
    o = buf.sample_batch(B, uniforms=u)
    e = ref.sample(u)
    assert np.array_equal(o[8].cpu().numpy(), e["idxes"])
    assert np.array_equal(o[0].float().cpu().numpy().astype(bool), e["obs"])
    assert np.array_equal(o[1].cpu().numpy()[:, 0], e["action"][:, 0])
    assert np.array_equal(o[2].cpu().numpy()[:, 0], e["reward"][:, 0].astype(np.float32))
    assert np.array_equal(o[3].cpu().numpy(), e["done"].astype(np.float32))
    assert np.array_equal(o[5].cpu().numpy(), e["bt_steps"])
    assert np.array_equal(o[6].float().cpu().numpy(), e["hidden"].astype(np.float32))
    assert np.array_equal(o[7].cpu().numpy(), e["comm_mask"])
    assert int(o[10]) == e["old_ptr"]


def test_outcome_log_feeds_level_table_in_order():
    """Curriculum statistics (worker.py:74-82) go through a device log that is read when statistics are asked for."""
    import mapf_rl_amd as M
    from mapf_rl_amd.actor import VecActor
    from mapf_rl_amd.model import Network
    from mapf_rl_amd.replay import GlobalBuffer

    torch.manual_seed(0)
    E, L, N = 32, 10, 1
    env = M.VecEnvironment(E, L, N)
    env.reset_envs(None, 0.1, seed=1)
    buf = GlobalBuffer(64, max_agents=6, init_set=(1, 10))
    actor = VecActor(env, Network().cuda().eval(), buf, epsilons=0.9, max_steps=6, seed=2, density=0.1, keep_flushed=True)
    for _ in range(30):
        actor.step()
    want = [ep["done"] for ep in actor.flushed if actor.actor_ids[ep["env"]] >= 10]
    assert buf.stat_dict[(1, 10)] == []      # nothing read yet
    buf.drain_outcomes()
    assert buf.stat_dict[(1, 10)] == want[-200:] and len(want) > 20
