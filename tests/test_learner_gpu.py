"""GPU (bf16 autocast): Learner.update against the reference golden (fp32): td error / loss within the bf16
tolerance, and the end-to-end device path replay.sample_batch -> update -> update_priorities."""
import os

import numpy as np
import pytest
import torch

from tests import helpers as H
from tests.test_learner_cpu import _batch, _models

pytestmark = pytest.mark.gpu


def test_update_bf16_vs_reference():
    z = H.load_npz("dqn_update.npz")
    lr = _models("cuda")
    out = lr.update(_batch(z, "cuda", torch.bfloat16))
    td, ref = out["td"].float().cpu().numpy(), z["td"]
    assert np.all(np.abs(td - ref) <= 4e-2 * np.maximum(1.0, np.abs(ref))), np.abs(td - ref).max()  # 2 bootstraps: 2 x 2e-2
    assert abs(float(out["loss"]) - float(z["loss"])) <= 5e-2 * max(1.0, float(z["loss"]))
    assert abs(float(out["grad_norm"]) - float(z["grad_norm"])) <= 0.1 * float(z["grad_norm"])


def test_device_pipeline_sample_update_priorities():
    from mapf_rl_amd.learner import Learner
    from mapf_rl_amd.replay import GlobalBuffer

    torch.manual_seed(0)
    rng = np.random.RandomState(0)
    A = 4
    buf = GlobalBuffer(8, max_agents=A)
    for k in range(8):
        size = int(rng.randint(20, 60))
        td = np.zeros(256)
        td[:size] = rng.random_sample(size) + 0.1
        buf.add_episode(A, rng.random_sample((size + 1, A, 6, 9, 9)) < 0.3, rng.randint(0, 5, size).astype(np.uint8),
                        rng.choice([-0.075, -0.5, 3.0], size).astype(np.float16), (rng.standard_normal((size, 256)) * 0.3).astype(np.float16),
                        td, bool(k % 2), size, rng.random_sample((size + 1, A, A)) < 0.5)
    lr = Learner(buf, device="cuda", batch_size=32)
    before = buf.priority_tree.tree().clone()
    losses = [float(lr.update()["loss"]) for _ in range(3)]
    after = buf.priority_tree.tree()
    assert all(np.isfinite(losses)) and lr.counter == 3
    assert not torch.equal(before, after)            # priorities were written back
    leaves = after[-buf.priority_tree.capacity:]
    assert abs(float(after[0]) - float(leaves.sum())) < 1e-6 * float(after[0])   # root == sum of leaves (buffer.py:30)


@pytest.mark.parametrize("double_q", [False, True])
def test_pipelined_target_forward_equals_sequential_order(double_q):
    """Learner(prefetch=True) samples batch k+1 and runs its target-network forward on a second stream while update k's backward
    is still in flight; the numbers must be those of the sequential order (same samples: the priorities of update k are
    written before batch k+1 is drawn; same target: nothing the backward does touches the target network).  With double_q the
    side stream also runs the ONLINE network (the arg-max) right after an optimizer step changed its weights: its packed weight
    images must have been rebuilt before the side stream reads them (Learner._update: model.prepack())."""
    from mapf_rl_amd.learner import Learner
    from mapf_rl_amd.model import Network
    from mapf_rl_amd.replay import GlobalBuffer

    from mapf_rl_amd.update import FusedUpdate

    def run(prefetch):
        torch.manual_seed(0)
        rng = np.random.RandomState(1)
        A = 5
        buf = GlobalBuffer(8, max_agents=A)
        for k in range(8):
            size = int(rng.randint(30, 70))
            td = np.zeros(256)
            td[:size] = rng.random_sample(size) + 0.1
            buf.add_episode(A, rng.random_sample((size + 1, A, 6, 9, 9)) < 0.3, rng.randint(0, 5, size).astype(np.uint8),
                            rng.choice([-0.075, -0.5, 3.0], size).astype(np.float16), (rng.standard_normal((size, 256)) * 0.3).astype(np.float16),
                            td, bool(k % 2), size, rng.random_sample((size + 1, A, A)) < 0.5)
        torch.manual_seed(7)
        torch.cuda.manual_seed(7)
        lr = Learner(buf, device="cuda", batch_size=24, model=Network(), prefetch=prefetch, double_q=double_q)
        assert lr.prefetch == prefetch and lr.double_q == double_q
        outs = [lr.update() for _ in range(5)]  # back to back: no actor step repacks the weights in between
        torch.cuda.synchronize()
        return outs, [p.detach().clone() for p in lr.model.parameters()], buf.priority_tree.tree().clone()

    graph, FusedUpdate.GRAPH = FusedUpdate.GRAPH, False  # (bit-for-bit: the graph replay pads its launches, test below)
    try:
        (o_seq, p_seq, t_seq), (o_pre, p_pre, t_pre) = run(False), run(True)
    finally:
        FusedUpdate.GRAPH = graph
    for a, b in zip(o_seq, o_pre):
        assert torch.equal(a["q_next"], b["q_next"]) and torch.equal(a["td"], b["td"])
        assert float(a["loss"]) == float(b["loss"])
    for a, b in zip(p_seq, p_pre):
        assert torch.equal(a, b)
    assert torch.equal(t_seq, t_pre)


def _fp32_module_q(net, batch, steps):
    """Network.bootstrap through the reference-shaped module path in fp32 on the CPU (no kernels, no autocast, no pruning)."""
    import copy

    cpu = copy.deepcopy(net).float().cpu()
    with torch.no_grad():
        return cpu.bootstrap(batch[0].float().cpu(), steps.cpu(), batch[6].float().cpu(), batch[7].cpu())


@pytest.mark.parametrize("tag", ["b40", "b128"])
def test_double_q_update_on_gpu(tag):
    """BASELINE config 5 names "double-DQN" (dead in the reference: config.py:46, worker.py:300-303 always takes max_a Q_target).
    Learner(double_q=True) on the reference-captured batches at 40 and 128 agents (tests/golden/dqn_big.npz), pruning on:
    q_next = (1 - done) Q_target(s', argmax_a Q_online(s', a)) against the plain formula through the fp32 module path (2e-2; where
    the online network's top two actions are closer than the tolerance either of them is a legitimate pick), td = q - (r +
    0.99^steps q_next) at 4e-2 (two bootstraps), and the third bootstrap -- the online network on the TARGET window's row plan --
    is the same function with and without pruning."""
    from mapf_rl_amd.learner import Learner
    from mapf_rl_amd.model import Network
    from tests import big_golden as BG
    from tests.test_learner_cpu import _models

    z = H.load_npz("dqn_big.npz")
    base = _models("cuda")
    lr = Learner(buffer=None, device="cuda", model=base.model, double_q=True)
    lr.tar_model.load_state_dict(base.tar_model.state_dict())
    b = BG.batch(z, tag, "cuda", torch.bfloat16)
    nxt = b[5] + b[4].view(-1).long()
    q_on = _fp32_module_q(lr.model, b, nxt).numpy().astype(np.float64)
    q_tar = _fp32_module_q(lr.tar_model, b, nxt).numpy().astype(np.float64)
    assert np.allclose(q_tar, z[tag + "_q_target_all"], rtol=1e-4, atol=1e-4)  # the fp32 path IS the reference's function
    done = b[3].float().cpu().numpy().astype(np.float64).reshape(-1)
    assert Network.PRUNE_UNREACHABLE
    got = lr.target_q(b).float().cpu().numpy().astype(np.float64).reshape(-1)
    tol = 2e-2

    def check_q_next(vals):
        for i in range(len(vals)):
            picks = np.nonzero(q_on[i] >= q_on[i].max() - 2 * tol * max(1.0, abs(q_on[i].max())))[0]
            cands = (1 - done[i]) * q_tar[i, picks]
            assert np.min(np.abs(vals[i] - cands)) <= tol * max(1.0, np.abs(cands).max()), (i, vals[i], cands)

    check_q_next(got)  # Learner.target_q: the autograd-level statement (Network.bootstrap x 2)
    # the same call with every observation encoded (the third bootstrap re-uses the target window's plan for the online network)
    try:
        Network.PRUNE_UNREACHABLE = False
        full = lr.target_q(b).float().cpu().numpy().astype(np.float64).reshape(-1)
    finally:
        Network.PRUNE_UNREACHABLE = True
    if tag == "b40":  # <= 48 agents: same bits (tests/test_relevance_gpu.py); above, the compacted recurrence: bf16 tolerance
        assert np.array_equal(got, full)
    else:
        assert np.all(np.abs(got - full) <= tol * np.maximum(1.0, np.abs(full)))
    # one full update: td against the plain formula with the fp32 online Q of the taken action
    bt = b[5]
    q_sel = _fp32_module_q(lr.model, (b[0][:, :-2],) + b[1:7] + (b[7][:, :-2],), bt).numpy().astype(np.float64)
    q_sel = np.take_along_axis(q_sel, b[1].cpu().numpy().reshape(-1, 1), axis=1).reshape(-1)
    rew, steps = b[2].float().cpu().numpy().reshape(-1).astype(np.float64), b[4].float().cpu().numpy().reshape(-1).astype(np.float64)
    out = lr.update(b)
    td = out["td"].float().cpu().numpy().astype(np.float64).reshape(-1)
    qn = out["q_next"].float().cpu().numpy().astype(np.float64).reshape(-1)
    check_q_next(qn)  # the update itself (update.FusedUpdate: third recurrence on the side stream, head in mapf_dqn_head_loss)
    # (and it agrees with target_q wherever the online arg-max is not a near tie: there the two paths may pick differently)
    clear = np.array([np.sum(q_on[i] >= q_on[i].max() - 2 * tol * max(1.0, abs(q_on[i].max()))) == 1 for i in range(len(qn))])
    assert np.all(np.abs(qn - got)[clear] <= tol * np.maximum(1.0, np.abs(got))[clear])
    want = q_sel - (rew + 0.99 ** steps * qn)  # the learner's own q_next: the pick is checked above
    assert np.all(np.isfinite(td)) and np.all(np.abs(td - want) <= 4e-2 * np.maximum(1.0, np.abs(want))), np.abs(td - want).max()
    assert np.isfinite(float(out["loss"])) and np.isfinite(float(out["grad_norm"])) and lr.counter == 1


def test_actors_on_their_own_stream_leave_the_replay_consistent():
    """train.py --overlap-actors: the actor iteration runs on its own stream beside the learner's update; its episode flush is
    ordered between two updates' replay operations by the events Learner.replay_released / replay_gate.  After a run the replay is
    what a serial run can produce: every sum-tree node is the sum of its children, priorities are finite and non-negative, the
    ring state counts what the actors flushed, and the update kept learning from finite numbers."""
    import mapf_rl_amd as M
    from mapf_rl_amd.actor import VecActor
    from mapf_rl_amd.learner import Learner
    from mapf_rl_amd.replay import GlobalBuffer

    torch.manual_seed(0)
    dev = torch.device("cuda")
    E, N, L = 96, 3, 12
    buf = GlobalBuffer(256, max_agents=N, device=dev, init_set=(N, L), fixed_level=True)
    lr = Learner(buf, device=dev, batch_size=32)
    env = M.VecEnvironment(E, L, N, device=dev)
    maps, agents, goals, _ = M.generate_scenarios(E, L, N, 0.2, seed=2)
    env.load(maps, agents, goals)
    actor = VecActor(env, lr.model, buf, seed=1, density=0.2, max_steps=12, weights_period=7)
    astream = torch.cuda.Stream(device=dev)
    for _ in range(30):  # fill
        actor.step()
    torch.cuda.synchronize()
    losses = []
    for it in range(60):
        if lr.replay_released is not None:
            astream.wait_event(lr.replay_released)
        with torch.cuda.stream(astream):
            actor.step()
            ev = torch.cuda.Event()
            ev.record(astream)
        lr.replay_gate = ev
        losses.append(lr.update()["loss"])
    torch.cuda.synchronize()
    assert all(bool(torch.isfinite(torch.as_tensor(v)).all()) for v in losses) and lr.counter == 60
    tree = buf.priority_tree.tree().cpu().numpy()
    cap = buf.priority_tree.capacity
    assert np.all(np.isfinite(tree)) and np.all(tree >= 0) and tree[0] > 0
    parents = tree[: cap - 1]
    assert np.allclose(parents, tree[1:2 * cap - 1:2] + tree[2:2 * cap - 1:2], rtol=1e-12, atol=0)
    ptr, size, counter, _ = buf.state()
    assert actor.episodes > 0 and 0 < size <= 256 * 256 and counter >= size
    assert all(bool(torch.isfinite(p).all()) for p in lr.model.parameters())


def test_overlapped_actors_equal_the_serial_order():
    """The same loop twice from the same seeds -- actors on their own stream beside the update (ordered only by the two events), and
    everything on one stream, strictly alternating: with the actors acting on a weight snapshot that is never refreshed inside the run
    (weights_period beyond its length) nothing they do depends on WHEN the update runs, so the episodes, the replay (ring state, every
    sum-tree node), the sampled batches and the parameters after 40 updates must be the SAME BITS.  A missing or misplaced
    dependency between the actors' episode flush and the learner's priority write-back / prioritized sample would show here."""
    import mapf_rl_amd as M
    from mapf_rl_amd.actor import VecActor
    from mapf_rl_amd.learner import Learner
    from mapf_rl_amd.model import Network
    from mapf_rl_amd.replay import GlobalBuffer

    def run(overlap):
        torch.manual_seed(0)
        torch.cuda.manual_seed(0)
        dev = torch.device("cuda")
        E, N, L = 96, 3, 12
        buf = GlobalBuffer(256, max_agents=N, device=dev, init_set=(N, L), fixed_level=True)
        lr = Learner(buf, device=dev, batch_size=32, model=Network())
        env = M.VecEnvironment(E, L, N, device=dev)
        maps, agents, goals, _ = M.generate_scenarios(E, L, N, 0.2, seed=2)
        env.load(maps, agents, goals)
        actor = VecActor(env, lr.model, buf, seed=1, density=0.2, max_steps=12, weights_period=10 ** 9)
        astream = torch.cuda.Stream(device=dev) if overlap else None
        for _ in range(30):
            actor.step()
        torch.cuda.synchronize()
        tds = []
        for it in range(40):
            if overlap:
                if lr.replay_released is not None:
                    astream.wait_event(lr.replay_released)
                with torch.cuda.stream(astream):
                    actor.step()
                    ev = torch.cuda.Event()
                    ev.record(astream)
                lr.replay_gate = ev
            else:
                actor.step()
            tds.append(lr.update()["td"].clone())
        torch.cuda.synchronize()
        return (torch.stack(tds), [p.detach().clone() for p in lr.model.parameters()], buf.priority_tree.tree().clone(), buf.state(), actor.episodes,
                actor.lb_obs.clone(), actor.t.clone())

    a, b = run(True), run(False)
    assert a[3] == b[3] and a[4] == b[4] and a[4] > 0
    assert torch.equal(a[5], b[5]) and torch.equal(a[6], b[6])
    assert torch.equal(a[2], b[2])
    assert torch.equal(a[0], b[0])
    for x, y in zip(a[1], b[1]):
        assert torch.equal(x, y)


def _filled_replay(A=6, episodes=24, seed=1):
    from mapf_rl_amd.replay import GlobalBuffer

    rng = np.random.RandomState(seed)
    buf = GlobalBuffer(32, max_agents=A)
    for k in range(episodes):
        size = int(rng.randint(30, 120))
        na = 1 + k % A                                     # levels of 1..A agents in one replay, like the curriculum
        td = np.zeros(256)
        td[:size] = rng.random_sample(size) + 0.1
        obs = rng.random_sample((size + 1, na, 6, 9, 9)) < 0.3
        obs[1::3] = obs[0:-1:3][:obs[1::3].shape[0]]       # repeated observations: the duplicate flags have something to find
        comm = rng.random_sample((size + 1, na, na)) < 0.4
        comm |= np.eye(na, dtype=bool)[None]
        buf.add_episode(na, obs, rng.randint(0, 5, size).astype(np.uint8), rng.choice([-0.075, -0.5, 3.0], size).astype(np.float16),
                        (rng.standard_normal((size, 256)) * 0.3).astype(np.float16), td, bool(k % 2), size, comm)
    return buf


@pytest.mark.parametrize("double_q", [False, True])
def test_graph_replayed_update_follows_the_eager_update(double_q):
    """update.FusedUpdate.GRAPH: at few agents the update is replayed from captured HIP graphs over bucket-sized (padded) launches.
    Same state, same samples (identical RNG, identical priorities to within the first update's rounding): the first update's
    Q-values / TD errors / loss / gradient norm and the parameters after it agree with the eager launch sequence to the tolerance of a
    different GEMM tiling (the padded rows contribute exact zeros); then a run of updates goes through replays of several buckets
    and stays close to the eager run."""
    from mapf_rl_amd.learner import Learner
    from mapf_rl_amd.model import Network
    from mapf_rl_amd.update import FusedUpdate

    def run(graph, n):
        saved, FusedUpdate.GRAPH = FusedUpdate.GRAPH, graph
        try:
            buf = _filled_replay()
            torch.manual_seed(7)
            torch.cuda.manual_seed(7)
            lr = Learner(buf, device="cuda", batch_size=48, model=Network(), double_q=double_q)
            outs = []
            for _ in range(n):
                o = lr.update()
                outs.append({k: v.detach().clone() for k, v in o.items()})
            torch.cuda.synchronize()
            fu = lr._fused
            return (outs, [p.detach().clone() for p in lr.model.parameters()], buf.priority_tree.tree().clone(), (fu.graph_replays, fu.graph_captures),
                    fu.flat.step, {k: fu.flat.mem(fu.flat.grads, k).clone() for k in fu.flat.names})
        finally:
            FusedUpdate.GRAPH = saved

    # one update: every parameter tensor's (clipped) gradient
    g_e, g_g = run(False, 1)[5], run(True, 1)[5]
    for k in g_e:
        err, ref = float((g_e[k] - g_g[k]).norm()), float(g_e[k].norm())
        assert err <= 2e-2 * ref + 1e-7, (k, err, ref)
    n = 14
    o_e, p_e, t_e, (rep_e, _), step_e, _ = run(False, n)
    o_g, p_g, t_g, (rep_g, cap_g), step_g, _ = run(True, n)
    assert rep_e == 0 and rep_g == n and 5 <= cap_g <= 6 * n and step_e == step_g == n
    a, b = o_e[0], o_g[0]
    for k in ("q", "q_next", "td"):
        assert torch.allclose(a[k], b[k], atol=2e-2, rtol=2e-2), (k, (a[k] - b[k]).abs().max())
    assert abs(float(a["loss"]) - float(b["loss"])) <= 2e-2 * max(1.0, abs(float(a["loss"])))
    assert abs(float(a["grad_norm"]) - float(b["grad_norm"])) <= 3e-2 * float(a["grad_norm"])
    assert torch.allclose(a["priorities"], b["priorities"], atol=2e-2, rtol=2e-2)
    # the whole run: same sampling stream, nearly the same priorities -> nearly the same batches; Adam steps are +-lr per element
    num = sum(float((x - y).pow(2).sum()) for x, y in zip(p_e, p_g))
    den = sum(float((x - p0).pow(2).sum()) for x, p0 in zip(p_e, run(False, 0)[1]))
    # the two runs moved the parameters the same way: difference < distance travelled.  A COARSE bound -- Adam turns every sign flip
    # of a near-zero gradient into +-lr and the sampled batches follow the priorities, so two EAGER runs that differ only in the
    # summation order of the encoder's weight gradients (20 instead of 21 partitions) already sit at 0.17 of the distance travelled,
    # where graph against eager reads 0.003 .. 0.17 depending on that order (tools/micro/graph_eager_drift.py,
    # profiles/r06_graph_eager_drift.txt); the per-tensor gradient check above is the sharp one, a broken stage gives >= 1 here
    assert num <= 0.5 * den, (num, den)
    for o in o_g:
        assert all(bool(torch.isfinite(v).all()) for v in o.values())
    leaves = t_g[-buf_leaves(t_g):]
    assert abs(float(t_g[0]) - float(leaves.sum())) < 1e-6 * float(t_g[0])


@pytest.mark.parametrize("dedup,prune,bounded", [(False, True, True), (True, False, True), (True, True, False)])
def test_graph_replayed_update_under_the_switches(dedup, prune, bounded):
    """The graph-replayed update with the observation reuse off (every reachable entry encoded: the bound of the bucket-sized encoder
    launches is then the entry total), with the pruning off (every entry of the window), and with the device-side row bound off (the
    kernels compute the whole bucket): the first update's gradients agree with the eager launch sequence tensor by tensor."""
    from mapf_rl_amd.learner import Learner
    from mapf_rl_amd.model import Network
    from mapf_rl_amd.update import FusedUpdate

    saved = (FusedUpdate.GRAPH, FusedUpdate.DEDUP, Network.PRUNE_UNREACHABLE, FusedUpdate.BOUNDED_ROWS)

    def run(graph):
        FusedUpdate.GRAPH, FusedUpdate.DEDUP, Network.PRUNE_UNREACHABLE, FusedUpdate.BOUNDED_ROWS = graph, dedup, prune, bounded
        buf = _filled_replay()
        torch.manual_seed(7)
        torch.cuda.manual_seed(7)
        lr = Learner(buf, device="cuda", batch_size=48, model=Network())
        outs = [{k: v.detach().clone() for k, v in lr.update().items()} for _ in range(3)]
        torch.cuda.synchronize()
        fu = lr._fused
        return outs, fu.graph_replays, {k: fu.flat.mem(fu.flat.grads, k).clone() for k in fu.flat.names}

    try:
        (o_e, rep_e, g_e), (o_g, rep_g, g_g) = run(False), run(True)
    finally:
        FusedUpdate.GRAPH, FusedUpdate.DEDUP, Network.PRUNE_UNREACHABLE, FusedUpdate.BOUNDED_ROWS = saved
    assert rep_e == 0 and rep_g == 3
    a, b = o_e[0], o_g[0]
    for k in ("q", "q_next", "td"):
        assert torch.allclose(a[k], b[k], atol=2e-2, rtol=2e-2), (k, (a[k] - b[k]).abs().max())
    assert abs(float(a["grad_norm"]) - float(b["grad_norm"])) <= 3e-2 * float(a["grad_norm"])
    for o in o_g:
        assert all(bool(torch.isfinite(v).all()) for v in o.values())
    for k in g_e:  # (the third update's gradients: nearly the same batches on both sides -- a coarse bound; finite everywhere)
        assert bool(torch.isfinite(g_g[k]).all()), k


def buf_leaves(tree):
    return (tree.numel() + 1) // 2


def _run_ranks(world, backend, n_updates, timeout=900):
    """Starts tests/dist_graph_worker.py as `world` ranks (all on GPU 0) and returns their JSON lines."""
    import json
    import socket
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(world):
        env = dict(os.environ, PYTHONPATH=root, RANK=str(r), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(root, "tests", "dist_graph_worker.py"), backend, str(n_updates)], env=env, cwd=root,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=timeout) for p in procs]
    assert [p.returncode for p in procs] == [0] * world, [o[1][-2500:] for o in outs]
    return [json.loads([l for l in o[0].splitlines() if l.startswith("{")][-1]) for o in outs]


def _check_rank_report(r, n):
    assert r["graph_mode_on"] and r["replays_eager"] == 0 and r["replays_graph"] == n and r["steps"] == [n, n] and r["finite"]
    # the exchange ran in BOTH launch sequences: two pieces and one wait per update
    assert r["exchange_calls"] == {"begin": 4 * n, "finish": 2 * n}, r["exchange_calls"]
    assert r["grad_err"] <= 2e-2, r                       # every parameter tensor's (exchanged, clipped) gradient: graph vs eager
    assert r["td_err"] <= 2e-2 and r["q_err"] <= 2e-2, r
    assert abs(r["loss"][0] - r["loss"][1]) <= 2e-2 * max(1.0, abs(r["loss"][0]))
    assert abs(r["grad_norm"][0] - r["grad_norm"][1]) <= 3e-2 * r["grad_norm"][0]
    assert r["param_diff_over_travel"] <= 0.1, r
    assert r["same_params_on_all_ranks"], r


def test_graph_replayed_update_with_the_exchange_inside_two_ranks():
    """Round-4 review: graph replay was switched off on more than one rank.  Two ranks (gloo, sharing GPU 0, each with its own
    episodes) run the eager launch sequence and the graph-replayed one with the gradient all-reduce inside: the graph path is on,
    the exchange is issued, both ranks end with bit-identical parameters, and replay follows eager within the tolerances of the
    single-rank test above."""
    n = 6
    for r in _run_ranks(2, "gloo", n):
        assert r["world"] == 2 and r["backend"] == "gloo"
        _check_rank_report(r, n)


def test_update_exchange_over_rccl_on_one_rank():
    """The collective calls themselves over RCCL (backend nccl): a ONE-rank group driven through the multi-rank code path
    (learner.FORCE_EXCHANGE) -- communicator creation, the two asynchronous all-reduce pieces issued from the side branch, the
    stream-ordered wait, graph stages with the exchange between them -- on the one GPU a test box has."""
    n = 6
    (r,) = _run_ranks(1, "nccl", n)
    assert r["world"] == 1 and r["backend"] == "nccl"
    _check_rank_report(r, n)
