#!/usr/bin/env python3
"""How far a graph-replayed run of updates drifts from the eager run (tests/test_learner_gpu.py::test_graph_replayed_update_follows_the_eager_update's
last check): sum |p_eager - p_graph|^2 / sum |p_eager - p_0|^2 after n updates, and the same between two EAGER runs that differ only in
the summation order of the encoder's weight gradients (MAPF_WGRAD_PARTS), i.e. the noise floor of the measure."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
from tests.test_learner_gpu import _filled_replay
from mapf_rl_amd.learner import Learner
from mapf_rl_amd.model import Network
from mapf_rl_amd.update import FusedUpdate

def run(graph, n):
    FusedUpdate.GRAPH = graph
    buf = _filled_replay()
    torch.manual_seed(7); torch.cuda.manual_seed(7)
    lr = Learner(buf, device="cuda", batch_size=48, model=Network())
    for _ in range(n):
        lr.update()
    torch.cuda.synchronize()
    return [p.detach().clone() for p in lr.model.parameters()]

n = int(sys.argv[1]) if len(sys.argv) > 1 else 14
p0 = run(False, 0)
pe = run(False, n)
pg = run(True, n)
pg2 = run(True, n)
den = sum(float((x - y).pow(2).sum()) for x, y in zip(pe, p0))
print("parts=%s merged=%s: graph vs eager %.4f   graph vs graph %.6f   (den %.4f)" % (os.environ.get("MAPF_WGRAD_PARTS"), os.environ.get("MAPF_WGRAD_MERGED"),
      sum(float((x - y).pow(2).sum()) for x, y in zip(pe, pg)) / den, sum(float((x - y).pow(2).sum()) for x, y in zip(pg, pg2)) / den, den))
torch.save(pe, "/tmp/drift_eager_%s_%s.pt" % (os.environ.get("MAPF_WGRAD_PARTS"), os.environ.get("MAPF_WGRAD_MERGED")))
