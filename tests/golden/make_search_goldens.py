#!/usr/bin/env python3
"""Generates tests/golden/search.npz from the UNMODIFIED reference planner (search.find_path = CBS with
disjoint splitting + space-time A*, reference search.py:58-442) on small seeded scenarios, plus
search.compute_heuristics distance fields.  The reference's CBS picks random conflicts and has a 5 s wall-clock
cut, so only scenario, makespan and sum of costs of ITS solution are recorded; the product's planner must be
valid and at most as costly."""
import os
import random
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref_harness as rh  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    ref = rh.load_reference()
    data = {}
    k = 0
    for (L, N, seed) in [(8, 3, 1), (10, 4, 2), (10, 6, 3), (12, 6, 4), (12, 8, 5), (16, 8, 6), (16, 10, 7), (20, 6, 8), (10, 1, 9), (14, 5, 10)]:
        np.random.seed(seed)
        random.seed(seed)
        env = None
        while env is None:
            try:
                env = ref.environment.Environment(map_length=L, num_agents=N)
            except ValueError:
                pass
        m, a, g = env.map.copy(), env.agents_pos.copy(), env.goals_pos.copy()
        random.seed(seed)
        actions = ref.search.find_path(env)
        if actions is None:
            continue
        acts = np.array(actions, np.int8).reshape(len(actions), N)
        cost = 0
        for i in range(N):
            nz = np.nonzero(acts[:, i])[0]
            cost += int(nz[-1]) + 1 if len(nz) else 0
        h = ref.search.compute_heuristics(m, tuple(int(v) for v in g[0]))
        dist = np.full((L, L), -1, np.int32)
        for (x, y), d in h.items():
            dist[x, y] = d
        pre = "case%d_" % k
        data[pre + "map"] = m.astype(np.int8)
        data[pre + "agents"] = a.astype(np.int8)
        data[pre + "goals"] = g.astype(np.int8)
        data[pre + "ref_actions"] = acts
        data[pre + "ref_cost"] = np.array(cost)
        data[pre + "dist0"] = dist
        print(pre, "L", L, "N", N, "makespan", len(actions), "sum of costs", cost)
        k += 1
    data["num_cases"] = np.array(k)
    np.savez_compressed(os.path.join(OUT, "search.npz"), **data)


if __name__ == "__main__":
    main()
