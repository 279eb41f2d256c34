"""Expert planner with the reference's `search.py` interface (reference search.py:24-55, 396-442):
`compute_heuristics(my_map, goal)` and `find_path(env)`, backed by the host C++ CBS in csrc/mapf_search.hip."""
import ctypes

import numpy as np

from . import _lib
from ._lib import check, lib

MAX_STEPS = 256  # config.max_steps (search.py:179)


def compute_heuristics(my_map, goal):
    """dict {(row, col): distance} over the cells reachable from `goal`, like reference search.py:24-55."""
    m = np.ascontiguousarray(np.asarray(my_map) != 0, dtype=np.int8)
    L = m.shape[0]
    dist = np.zeros((L, L), np.int32)
    check(lib.mapf_distance_field(L, m.ctypes.data, int(goal[0]), int(goal[1]), dist.ctypes.data), "mapf_distance_field")
    xs, ys = np.nonzero(dist >= 0)
    return {(int(x), int(y)): int(dist[x, y]) for x, y in zip(xs, ys)}


def plan(map_, agents_pos, goals_pos, time_limit=5.0, max_steps=MAX_STEPS):
    """Returns (actions int8 [T, N], sum_of_costs) or None when no plan was found within `time_limit` seconds."""
    m = np.ascontiguousarray(np.asarray(map_) != 0, dtype=np.int8)
    a = np.ascontiguousarray(agents_pos, dtype=np.int16)
    g = np.ascontiguousarray(goals_pos, dtype=np.int16)
    L, N = m.shape[0], a.shape[0]
    out = np.zeros((max_steps, N), np.int8)
    steps, cost = ctypes.c_int(0), ctypes.c_int(0)
    st = lib.mapf_find_path(L, N, m.ctypes.data, a.ctypes.data, g.ctypes.data, float(time_limit), int(max_steps),
                            out.ctypes.data, ctypes.byref(steps), ctypes.byref(cost))
    if st == _lib.ERR_TIMEOUT:
        return None
    if st == _lib.ERR_NO_SPACE:
        raise AssertionError("no solution for A-star search")  # reference search.py:334
    check(st, "mapf_find_path")
    return out[:steps.value].copy(), cost.value


def find_path(env, time_limit=5.0):
    """reference search.py:396-442: list of per-step action lists (plain ints for a single agent) or None."""
    res = plan(env.map, env.agents_pos, env.goals_pos, time_limit)
    if res is None:
        return None
    actions, _ = res
    if env.num_agents == 1:
        return [int(r[0]) for r in actions]
    return [[int(v) for v in r] for r in actions]
