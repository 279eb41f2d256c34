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
