"""CPU: the curriculum's level table (reference worker.py:74-82 statistics, :205-226 promotion, :237-250 stop) and its
multi-rank form: pooled windows give every rank the same decisions (2 gloo ranks with different local outcomes)."""
import torch.multiprocessing as mp
import os
import socket
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_promotion_rule_matches_reference():
    from mapf_rl_amd.curriculum import LevelTable

    t = LevelTable((1, 10), max_agents=6, max_map_length=40, pass_rate=0.9)
    for i in range(199):
        t.record((1, 10), True)
    assert t.advance() == ["(1, 10): 199/199"] and t.levels == [(1, 10)]  # 200 results are needed (worker.py:211)
    t.record((1, 10), True)
    t.record((7, 7), True)  # unknown level: ignored (worker.py:76)
    t.advance()
    assert sorted(t.levels) == [(1, 15), (2, 10)]  # +1 agent, +5 map side, old level retired
    t.windows[(2, 10)] = [True] * 179 + [False] * 21  # 89.5 % is not enough
    t.advance()
    assert (2, 10) in t.levels and (3, 10) not in t.levels
    # sliding window of 200
    for _ in range(300):
        t.record((2, 10), True)
    assert t.counts()[(2, 10)] == (200, 200)
    # caps: agent count at max_agents, map at max_map_length; the longest map is never retired (worker.py:214-222)
    t.windows = {(6, 40): [True] * 200}
    t.advance()
    assert t.levels == [(6, 40)] and not t.done()
    for n in range(1, 7):
        t.windows[(n, 40)] = [True] * 200
    assert t.done()
    t.windows[(3, 40)][0] = False
    t.windows[(3, 40)] = t.windows[(3, 40)] * 1
    assert t.done()  # 199/200 >= 90 %
    t.windows[(3, 40)] = [True] * 150
    assert not t.done()


def test_fixed_level_never_promotes():
    from mapf_rl_amd.curriculum import LevelTable

    t = LevelTable((40, 32), max_agents=40, fixed=True)
    for _ in range(250):
        t.record((40, 32), True)
    assert t.advance() == ["(40, 32): 200/200"] and t.levels == [(40, 32)]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank_main(rank, world, port, root, ret):
    sys.path.insert(0, root)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mapf_rl_amd.curriculum import LevelTable

    t = LevelTable((1, 10))
    # rank 0 alone would promote (200/200), rank 1 alone would not (150/200): pooled 350/400 = 87.5 % -> nobody promotes
    for i in range(200):
        t.record((1, 10), rank == 0 or i < 150)
    pooled = t.pooled_counts(torch.device("cpu"))
    t.advance(pooled, t.WINDOW * world)
    first = (dict(pooled), list(t.levels))
    # now rank 1 catches up: pooled 390/400 -> both promote, to the same level set
    for i in range(200):
        t.record((1, 10), rank == 0 or i < 190)
    pooled = t.pooled_counts(torch.device("cpu"))
    t.advance(pooled, t.WINDOW * world)
    ret[rank] = (first, dict(pooled), sorted(t.levels), t.done(t.pooled_counts(torch.device("cpu")), t.WINDOW * world))
    dist.destroy_process_group()


def test_two_ranks_take_identical_decisions():
    mgr = mp.get_context("spawn").Manager()
    ret = mgr.dict()
    mp.spawn(_rank_main, args=(2, _free_port(), ROOT, ret), nprocs=2, join=True)
    assert ret[0] == ret[1]
    first, pooled, levels, done = ret[0]
    assert first[0][(1, 10)] == (350, 400) and first[1] == [(1, 10)]
    assert pooled[(1, 10)] == (390, 400) and levels == [(1, 15), (2, 10)] and done is False
