"""Shared by tests/test_generator_stats.py (CPU, mapf_generate) and tests/test_reset_gpu.py (GPU, reset_kernel): the
statistics of tests/golden/gen_stats.npz (captured from the reference's own generator by tests/golden/make_gen_stats.py)
for a batch of scenarios, and the two-sample comparison.

Bound: two-sample Kolmogorov-Smirnov distance <= 0.05 for every statistic (reference sample: 2,000 scenarios per shape;
ours: E scenarios; at n = 2,000, m >= 4,000 a KS distance of 0.05 is the alpha ~ 0.002 critical value, so the bound holds for
samples of one distribution and fails for a placement rule that is visibly off: uniform-over-cells placement without the
partition weighting moves `in_largest`, a goal drawn from another component makes `dist` infinite, a different density law
moves `rho` by > 0.1)."""
import numpy as np
from scipy import ndimage, stats

from oracle import oracle
from tests import helpers as H

KS_BOUND = 0.05
FIELDS = ("rho", "ncomp", "largest_frac", "dist", "manhattan")


def batch_stats(maps, agents, goals):
    out = {k: [] for k in FIELDS + ("in_largest",)}
    for m, a, g in zip(np.asarray(maps), np.asarray(agents), np.asarray(goals)):
        free = m == 0
        lab, n = ndimage.label(free)  # 4-connectivity, like map_partition (environment.py:21-70)
        sizes = np.bincount(lab.ravel(), minlength=n + 1)[1:]
        big = int(np.argmax(sizes)) + 1
        out["rho"].append(float(m.mean()))
        out["ncomp"].append(int((sizes >= 2).sum()))
        out["largest_frac"].append(float(sizes.max() / free.sum()))
        for p, q in zip(a, g):
            dm = oracle.dist(m.astype(np.int8), (int(q[0]), int(q[1])))
            out["dist"].append(int(dm[int(p[0]), int(p[1])]))
            out["manhattan"].append(abs(int(p[0]) - int(q[0])) + abs(int(p[1]) - int(q[1])))
            out["in_largest"].append(int(lab[int(p[0]), int(p[1])] == big))
    return {k: np.asarray(v) for k, v in out.items()}


def compare(N, L, ours):
    """Returns {statistic: KS distance} (+ the in_largest mean difference); asserts nothing."""
    z = H.load_npz("gen_stats.npz")
    pre = "n%d_l%d_" % (N, L)
    res = {k: float(stats.ks_2samp(z[pre + k], ours[k]).statistic) for k in FIELDS}
    res["in_largest_diff"] = abs(float(z[pre + "in_largest"].mean()) - float(ours["in_largest"].mean()))
    res["ref_failure_rate"] = float(z[pre + "failures"]) / float(z["S"])
    return res


def check(N, L, ours):
    res = compare(N, L, ours)
    assert ours["dist"].max() < 2147483647, "a goal is unreachable from its start"
    for k in FIELDS:
        assert res[k] <= KS_BOUND, (k, res)
    assert res["in_largest_diff"] <= 0.02, res
    return res
