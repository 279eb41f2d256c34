#!/usr/bin/env python3
"""Actor iteration over several active curriculum levels: all levels stepped together (CurriculumActors.BATCHED: one change detection,
one encoder launch, one projection GEMM, one Q head) against the levels stepped one after the other (round 2's loop), with and without
reuse of unchanged observations' latents.  Usage: curriculum_iter.py [envs_per_level] [iterations]   (rocprofv3 --kernel-trace counts the launches)"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from mapf_rl_amd.actor import VecActor
from mapf_rl_amd.curriculum import CurriculumActors
from mapf_rl_amd.model import Network
from mapf_rl_amd.replay import GlobalBuffer
E = int(sys.argv[1]) if len(sys.argv) > 1 else 512
K = int(sys.argv[2]) if len(sys.argv) > 2 else 60
levels = [(4, 15), (3, 20), (2, 25), (6, 15), (5, 20), (1, 30), (4, 25)]  # the level set of profiles/r02_train_curriculum_5min.log's last interval
MODES = ((True, True, True, True), (True, True, True, False), (True, True, False, False), (True, False, False, False), (False, False, False, False))
if os.environ.get("MODES") == "graph":  # (profiling runs: the graph-replayed iteration only, last in the trace)
    MODES = MODES[:1]
for batched, reuse, merged, graph in MODES:
    CurriculumActors.BATCHED, VecActor.REUSE_LATENTS, CurriculumActors.MERGED, CurriculumActors.GRAPH = batched, reuse, merged, graph
    torch.manual_seed(0)
    net = Network().cuda().eval()
    buf = GlobalBuffer(16384, max_agents=6, init_set=(1, 10), max_map_length=40, pass_rate=0.9)
    buf.stat_dict = {k: [] for k in levels}
    cur = CurriculumActors(net, buf, envs_per_level=E, seed=3)
    for _ in range(40):
        cur.step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(K):
        cur.step()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    steps = len(levels) * E
    print("%d levels x %d envs, batched=%s reuse=%s merged env launches=%s graph=%s: %.3f ms per iteration (host enqueue %.3f ms) = %.3g env-steps/s" % (
        len(levels), E, batched, reuse, merged, graph, (t2 - t0) / K * 1e3, (t1 - t0) / K * 1e3, steps / ((t2 - t0) / K)), flush=True)
CurriculumActors.BATCHED, VecActor.REUSE_LATENTS, CurriculumActors.MERGED, CurriculumActors.GRAPH = True, True, True, True
