"""GPU: curriculum of (num_agents, map_length) levels (reference worker.py:74-82,205-250): level promotion
rule, actors following the level list, mixed agent counts in one replay."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_level_promotion_and_mixed_levels():
    from mapf_rl_amd.curriculum import CurriculumActors
    from mapf_rl_amd.model import Network
    from mapf_rl_amd.replay import GlobalBuffer

    torch.manual_seed(0)
    buf = GlobalBuffer(64, max_agents=6, init_set=(1, 10), max_map_length=40, pass_rate=0.9)
    net = Network().cuda().eval()
    cur = CurriculumActors(net, buf, envs_per_level=16, seed=1, max_steps=16)
    assert list(cur.actors) == [(1, 10)]
    for _ in range(20):
        cur.step()
    assert len(buf) > 0 and cur.episodes >= 16
    # not enough statistics yet: 200 results are needed (worker.py:211)
    buf.stats(1.0)
    assert cur.sync_levels() == [(1, 10)]
    # force the pass criterion: 200 results with >= 90 % success -> +1 agent and +5 map side, old level retired
    buf.stat_dict[(1, 10)] = [True] * 185 + [False] * 15
    buf.stats(1.0)
    assert sorted(buf.get_level()) == [(1, 15), (2, 10)]
    assert sorted(cur.sync_levels()) == [(1, 15), (2, 10)] and sorted(cur.actors) == [(1, 15), (2, 10)]
    before = len(buf)
    for _ in range(20):
        cur.step()
    assert len(buf) > before
    # 89.5 % is not enough
    buf.stat_dict[(2, 10)] = [True] * 179 + [False] * 21
    buf.stats(1.0)
    assert (2, 10) in buf.get_level() and (3, 10) not in buf.get_level()
    # the promotion rule alone, as train.py --promote-interval applies it between two statistics: same decisions, nothing printed
    buf.stat_dict[(1, 15)] = [True] * 190 + [False] * 10
    import contextlib
    import io

    with contextlib.redirect_stdout(io.StringIO()) as said:
        lines = buf.advance_levels()
    assert said.getvalue() == "" and any(l.startswith("(1, 15): 190/200") for l in lines)
    assert (1, 15) not in buf.get_level() and (1, 20) in buf.get_level() and (2, 15) in buf.get_level()
    assert sorted(cur.sync_levels()) == sorted(buf.get_level())
    for _ in range(5):
        cur.step()
    # the agent count is capped at max_num_agetns and the map at max_map_lenght (worker.py:214,217)
    buf.stat_dict = {(6, 40): [True] * 200}
    buf.stats(1.0)
    assert buf.get_level() == [(6, 40)] and not buf.check_done()
    for i in range(6):
        buf.stat_dict[(i + 1, 40)] = [True] * 200
    assert buf.check_done()
    # mixed agent counts in one ring: windows of 1- and 2-agent episodes come back padded to 6 agents
    out = buf.sample_batch(32)
    obs = out[0].float()
    assert obs.shape == (32, 18, 6, 6, 9, 9)
    assert float(obs[:, :, 2:].abs().sum()) == 0.0          # no level had more than 2 agents
    assert float(obs[:, :, 0].abs().sum()) > 0


@pytest.mark.parametrize("mode", ["batched", "merged", "graph"])
def test_level_batched_step_equals_serial_steps(mode):
    """CurriculumActors.step with all active levels through ONE change detection / encoder launch / projection GEMM / Q head
    (Network.step_levels) against the levels stepped one after the other (BATCHED = False), from the same seeds: every recorded
    row -- observations, actions, rewards, comm rows, episode sizes, the replay's ring state and sum tree -- is identical; Q-values
    and hidden states agree to bf16 rounding (the GEMM runs on another row count).
    mode "merged": also the levels' environment step / reset / re-observation as ONE launch each (mapf_multi_*, per-level scenario
    and exploration streams kept: same episodes); mode "graph": and the whole iteration replayed from a captured HIP graph (the
    iteration counter that moves those streams lives on the device)."""
    from mapf_rl_amd.curriculum import CurriculumActors
    from mapf_rl_amd.model import Network
    from mapf_rl_amd.replay import GlobalBuffer

    levels = [(1, 10), (2, 10), (3, 15), (6, 20), (4, 25), (5, 10), (6, 35), (2, 40)]
    runs = {}
    saved = (CurriculumActors.BATCHED, CurriculumActors.MERGED, CurriculumActors.GRAPH)
    try:
        for batched in (True, False):
            CurriculumActors.BATCHED, CurriculumActors.MERGED, CurriculumActors.GRAPH = batched, mode in ("merged", "graph"), mode == "graph"
            torch.manual_seed(0)
            net = Network().cuda().eval()
            buf = GlobalBuffer(1024, max_agents=6, init_set=(1, 10), max_map_length=40, pass_rate=0.9)
            buf.stat_dict = {k: [] for k in levels}
            cur = CurriculumActors(net, buf, envs_per_level=32, seed=3, max_steps=12)
            assert sorted(cur.actors) == sorted(levels)
            assert (cur.multi is not None) == (batched and mode != "batched")
            for _ in range(30):
                cur.step()
            torch.cuda.synchronize()
            for a in cur.actors.values():
                a.env.check_status()
            rec = {k: dict(obs=a.lb_obs.clone(), act=a.lb_act.clone(), rew=a.lb_rew.clone(), comm=a.lb_comm.clone(), t=a.t.clone(), q=a.lb_q.clone(),
                           hid=a.lb_hid.clone(), pos=a.pos.clone(), eps=a.episodes, steps=a.env_steps) for k, a in cur.actors.items()}
            runs[batched] = (rec, buf.state(), buf.priority_tree.tree().clone(), cur.latents, cur.graph_replays)
    finally:
        CurriculumActors.BATCHED, CurriculumActors.MERGED, CurriculumActors.GRAPH = saved
    (ra, sa, ta, lat, replays), (rb, sb, tb, _, _) = runs[True], runs[False]
    assert lat is not None and lat.full == 1  # one cache over all levels; re-encoded in full only at the start
    assert replays == (28 if mode == "graph" else 0) and lat.calls == (3 if mode == "graph" else 30)  # (2 direct iterations + the capture)
    assert sa == sb and sa[2] > 0
    for k in levels:
        for f in ("obs", "act", "rew", "comm", "t", "pos"):
            assert torch.equal(ra[k][f], rb[k][f]), (k, f)
        assert ra[k]["eps"] == rb[k]["eps"] and ra[k]["eps"] >= 32 and ra[k]["steps"] == rb[k]["steps"] == 30 * 32
        assert torch.allclose(ra[k]["q"], rb[k]["q"], rtol=2e-2, atol=2e-2) and torch.allclose(ra[k]["hid"].float(), rb[k]["hid"].float(), rtol=2e-2, atol=2e-2)
    assert torch.allclose(ta, tb, rtol=1e-2, atol=1e-6)  # priorities are |td| of those Q-values


@pytest.mark.parametrize("update_graph", [False, True])
def test_graph_replayed_actor_iterations_interleaved_with_learner_updates(update_graph):
    """train.py's loop in small: the curriculum actors' iteration replayed from its HIP graph, learner updates in between (issued
    directly, or replayed from the update's own graphs).  Regression for a runtime limit found in round 4: a captured graph that
    holds a small hipMemsetAsync NODE (mapf_obs_changed zeroed its 4-byte row counter that way) faulted at a later replay once the
    learner's launches -- which issue hipMemsetAsync themselves -- had run in between (tools/micro/graph_gemm_probe.py).  Nothing
    that can be captured calls hipMemsetAsync any more (tiny kernels / fill kernels instead)."""
    import config
    from mapf_rl_amd.curriculum import CurriculumActors
    from mapf_rl_amd.learner import Learner
    from mapf_rl_amd.replay import GlobalBuffer
    from mapf_rl_amd.update import FusedUpdate

    saved, FusedUpdate.GRAPH = FusedUpdate.GRAPH, update_graph
    try:
        torch.manual_seed(0)
        levels = [(1, 10), (3, 15), (6, 20)]
        buf = GlobalBuffer(2048, max_agents=6, init_set=(1, 10), max_map_length=40, pass_rate=0.9)
        buf.stat_dict = {k: [] for k in levels}
        lr = Learner(buf, device="cuda", batch_size=64)
        cur = CurriculumActors(lr.model, buf, envs_per_level=128, seed=0, max_steps=24, reward_fn=config.reward_fn, weights_period=40)
        for _ in range(60):
            cur.step()
        assert cur.graph_replays >= 55 and len(buf) >= 64 * 18
        for rnd in range(30):
            for _ in range(2):
                out = lr.update()
            for _ in range(5):
                cur.step()
        torch.cuda.synchronize()
        for a in cur.actors.values():
            a.env.check_status()
        assert bool(torch.isfinite(out["loss"])) and lr.counter == 60
        assert lr._fused.graph_replays == (60 if update_graph else 0)
        tree = buf.priority_tree.tree()
        leaves = tree[-buf.priority_tree.capacity:]
        assert abs(float(tree[0]) - float(leaves.sum())) < 1e-6 * float(tree[0])
    finally:
        FusedUpdate.GRAPH = saved


def test_actors_on_the_learners_own_module_leave_the_graph_when_its_weights_change():
    """Advisor, round 4: with weights_period=None the actors act on the learner's module itself; a captured iteration reads the packed
    weight images and cached latents of the last directly issued iteration, so replaying it after an optimizer step would mix stale
    encoder / recurrence weights with the live head.  The graph is keyed on the acting weights (epoch + every parameter's address and
    version): while they stand still iterations are replayed, after any change one is issued directly, re-packing the images and
    re-encoding every observation."""
    import config
    from mapf_rl_amd.curriculum import CurriculumActors
    from mapf_rl_amd.model import Network
    from mapf_rl_amd.replay import GlobalBuffer

    def make(model):
        buf = GlobalBuffer(512, max_agents=6, init_set=(1, 10), max_map_length=40, pass_rate=0.9)
        buf.stat_dict = {k: [] for k in [(2, 10), (4, 15)]}
        return CurriculumActors(model, buf, envs_per_level=64, seed=3, max_steps=24, reward_fn=config.reward_fn, weights_period=None)

    torch.manual_seed(1)
    model = Network().cuda()
    cur = make(model)
    for _ in range(8):
        cur.step()
    r0 = cur.graph_replays
    assert r0 >= 4
    with torch.no_grad():                       # "an optimizer step": every parameter moves, in place
        for p in model.parameters():
            p.add_(0.01 * torch.randn_like(p))
    model.weights_epoch += 1
    cur.step()
    assert cur.graph_replays == r0              # issued directly, not replayed
    rows = sum(a.E * a.N for a in cur.actors.values())
    assert cur.latents.last_encoded() == rows   # ... with every observation re-encoded under the new weights (no stale latent survives)
    cur.step()
    assert cur.graph_replays == r0 + 1          # replayed again from then on
    torch.cuda.synchronize()
    for a in cur.actors.values():
        a.env.check_status()
