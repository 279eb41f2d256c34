#!/usr/bin/env python3
"""Generates tests/golden/gen_stats.npz: statistics of the UNMODIFIED reference's scenario generator
(`Environment.__init__`, reference environment.py:75-144: obstacle density ~ triangular(0, 0.33, 0.5), Bernoulli map,
`map_partition` :21-70, partition-weighted start/goal placement :118-138), run through oracle/ref_harness.py.
Build container only; the output is data (per-scenario / per-agent samples of scalar statistics).

Per shape (num_agents, map_length) in SHAPES, S seeded scenarios (numpy + random seeded with 100000*k + s):
  rho            f32 [S]    realised obstacle fraction of the map
  ncomp          i16 [S]    number of free-cell components with >= 2 cells (the candidates of :105)
  largest_frac   f32 [S]    share of the free cells in the largest component
  dist           i16 [S*N]  BFS (shortest-path) distance start -> goal per agent
  manhattan      i16 [S*N]  |dx| + |dy| start -> goal per agent
  in_largest     u8  [S*N]  agent placed in the largest component
  failures       i32        constructions that raised (placement ran out of cells, :120 ValueError)
The navi map (:142, 0.15-0.8 s per scenario) is not part of the scenario: the harness replaces the method on the imported
class by a no-op for this run (the file on disk is untouched).
"""
import multiprocessing as mp
import os
import random
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
OUT = os.path.dirname(os.path.abspath(__file__))
SHAPES = [(6, 20), (40, 32), (16, 40)]
S = 2000


def scenario_stats(m, agents, goals):
    """Shared by the tests: the same statistics for any (map, agents, goals)."""
    from scipy import ndimage

    from oracle import oracle

    m = np.asarray(m)
    L = m.shape[0]
    free = m == 0
    lab, n = ndimage.label(free)  # 4-connectivity
    sizes = np.bincount(lab.ravel(), minlength=n + 1)[1:]
    big = int(np.argmax(sizes)) + 1 if n else 0
    d, mh, il = [], [], []
    for a, g in zip(np.asarray(agents), np.asarray(goals)):
        dm = oracle.dist(m.astype(np.int8), (int(g[0]), int(g[1])))
        d.append(int(dm[int(a[0]), int(a[1])]))
        mh.append(abs(int(a[0]) - int(g[0])) + abs(int(a[1]) - int(g[1])))
        il.append(int(lab[int(a[0]), int(a[1])] == big))
    return dict(rho=float(m.mean()), ncomp=int((sizes >= 2).sum()), largest_frac=float(sizes.max() / max(1, free.sum())) if n else 0.0,
                dist=d, manhattan=mh, in_largest=il)


def _work(args):
    k, (N, L), lo, hi = args
    from oracle import ref_harness as rh

    ref = rh.load_reference()
    ref.environment.Environment.get_navi_map = lambda self: None  # harness-side: see module docstring
    out = []
    fails = 0
    for s in range(lo, hi):
        np.random.seed(100000 * k + s)
        random.seed(100000 * k + s)
        try:
            env = ref.environment.Environment(num_agents=N, map_length=L)
        except ValueError:
            fails += 1
            continue
        out.append(scenario_stats(np.asarray(env.map), env.agents_pos, env.goals_pos))
    return k, out, fails


def main():
    jobs = []
    chunk = 50
    for k, shape in enumerate(SHAPES):
        for lo in range(0, S, chunk):
            jobs.append((k, shape, lo, min(S, lo + chunk)))
    res = {k: ([], 0) for k in range(len(SHAPES))}
    with mp.Pool(8) as pool:
        for k, out, fails in pool.imap_unordered(_work, jobs):
            res[k] = (res[k][0] + out, res[k][1] + fails)
            print("shape", SHAPES[k], "scenarios", len(res[k][0]), flush=True)
    data = {"shapes": np.array(SHAPES), "S": np.array(S)}
    for k, (N, L) in enumerate(SHAPES):
        out, fails = res[k]
        pre = "n%d_l%d_" % (N, L)
        data[pre + "rho"] = np.array([o["rho"] for o in out], np.float32)
        data[pre + "ncomp"] = np.array([o["ncomp"] for o in out], np.int16)
        data[pre + "largest_frac"] = np.array([o["largest_frac"] for o in out], np.float32)
        data[pre + "dist"] = np.array([v for o in out for v in o["dist"]], np.int16)
        data[pre + "manhattan"] = np.array([v for o in out for v in o["manhattan"]], np.int16)
        data[pre + "in_largest"] = np.array([v for o in out for v in o["in_largest"]], np.uint8)
        data[pre + "failures"] = np.array(fails, np.int32)
        print(pre, "ok", len(out), "failures", fails, "rho mean %.4f" % data[pre + "rho"].mean(), "dist mean %.2f" % data[pre + "dist"].mean(),
              "manhattan mean %.2f" % data[pre + "manhattan"].mean())
    np.savez_compressed(os.path.join(OUT, "gen_stats.npz"), **data)


if __name__ == "__main__":
    main()
