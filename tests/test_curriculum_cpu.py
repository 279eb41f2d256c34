"""CPU: the curriculum's level table (reference worker.py:74-82 statistics, :205-226 promotion, :237-250 stop) and its
multi-rank form: pooled windows give every rank the same decisions (2 gloo ranks with different local outcomes)."""
import torch.multiprocessing as mp
import os
import socket
import sys

import pytest
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

    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mapf_rl_amd.curriculum import LevelTable

    t = LevelTable((1, 10))
    # the even ranks alone would promote (200/200), the odd ranks alone would not (150/200): pooled 87.5 % -> nobody promotes
    for i in range(200):
        t.record((1, 10), rank % 2 == 0 or i < 150)
    pooled = t.pooled_counts(torch.device("cpu"))
    t.advance(pooled, t.WINDOW * world)
    first = (dict(pooled), list(t.levels))
    # now the odd ranks catch up: pooled 97.5 % -> every rank promotes, to the same level set
    for i in range(200):
        t.record((1, 10), rank % 2 == 0 or i < 190)
    pooled = t.pooled_counts(torch.device("cpu"))
    t.advance(pooled, t.WINDOW * world)
    # the control plane of train.py: three flags MAX-reduced asynchronously on host tensors, read one decision period later
    mine = torch.tensor([int(rank == world - 1), 0, int(rank == 0)], dtype=torch.int32)
    work = dist.all_reduce(mine, op=dist.ReduceOp.MAX, async_op=True)
    work.wait()
    ret[rank] = (first, dict(pooled), sorted(t.levels), t.done(t.pooled_counts(torch.device("cpu")), t.WINDOW * world), mine.tolist())
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_n_ranks_take_identical_decisions(world):
    """2 ranks, and 8 (the node's size, BASELINE configs[3]): pooled level windows and reduced control flags give every rank the same
    promotion / start / stop decisions whatever its own episodes say."""
    mgr = mp.get_context("spawn").Manager()
    ret = mgr.dict()
    mp.spawn(_rank_main, args=(world, _free_port(), ROOT, ret), nprocs=world, join=True)
    assert all(ret[r] == ret[0] for r in range(world))
    first, pooled, levels, done, flags = ret[0]
    h = world // 2
    assert first[0][(1, 10)] == (350 * h, 400 * h) and first[1] == [(1, 10)]
    assert pooled[(1, 10)] == (390 * h, 400 * h) and levels == [(1, 15), (2, 10)] and done is False
    assert flags == [1, 0, 1]
