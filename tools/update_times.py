#!/usr/bin/env python3
"""Host-enqueue time vs wall time of one learner update on replay windows produced by the actor loop (so that the communication
masks -- and with them the share of observations that can reach agent 0's Q-value -- are real), with and without the pruning of
unreachable observations.  Usage: update_times.py [agents] [map] [envs]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mapf_rl_amd as M  # noqa: E402
from mapf_rl_amd.actor import VecActor  # noqa: E402
from mapf_rl_amd.learner import Learner  # noqa: E402
from mapf_rl_amd.model import Network, relevance  # noqa: E402
from mapf_rl_amd.replay import GlobalBuffer  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
L = int(sys.argv[2]) if len(sys.argv) > 2 else 32
E = int(sys.argv[3]) if len(sys.argv) > 3 else 2048
dev = torch.device("cuda")
torch.manual_seed(0)
buf = GlobalBuffer(4096, max_agents=max(N, 6), device=dev, init_set=(N, L), fixed_level=True)
lr = Learner(buf, device=dev, batch_size=192)
env = M.VecEnvironment(E, L, N, device=dev)
maps, agents, goals, _ = M.generate_scenarios(E, L, N, 0.3, seed=1)
env.load(maps, agents, goals)
from bench import heuristic_actions  # noqa: E402  (the executed actions while the replay fills: 80 % heuristic-following, so that the agents move)
hgen = torch.Generator(device=dev).manual_seed(11)
actor = VecActor(env, lr.model, buf, seed=0)
for _ in range(300):
    actor.step(actions_override=heuristic_actions(actor.obs, hgen).long())
b = buf.sample_batch(192)
rel = relevance(b[7][:, :-2], b[5])
per_window = rel[0].sum(dim=1)
print("%d agents, %dx%d: %.3f of the online window's (step, agent) entries can reach agent 0; agents that matter per window: mean %.1f, max %d" % (
    N, L, L, float(rel.float().mean()), float(per_window.float().mean()), int(per_window.max())))
print("windows by agents that matter: <=16: %d, 17..32: %d, 33..48: %d, >48: %d of %d" % (
    int((per_window <= 16).sum()), int(((per_window > 16) & (per_window <= 32)).sum()), int(((per_window > 32) & (per_window <= 48)).sum()),
    int((per_window > 48).sum()), per_window.numel()), flush=True)
from mapf_rl_amd.update import FusedUpdate  # noqa: E402

ITERS = int(os.environ.get("ITERS", "100"))
for graph in (False, True):
    FusedUpdate.GRAPH = graph
    for prune in (False, True):
        Network.PRUNE_UNREACHABLE = prune
        lr._drop_prefetch()
        for _ in range(int(os.environ.get("WARM", "40")) if graph else 3):  # (graph mode: the buckets this replay produces get captured)
            lr.update()
        torch.cuda.synchronize()
        c0 = lr._fused.graph_captures
        t0 = time.perf_counter()
        for _ in range(ITERS):
            lr.update()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print("graph=%-5s prune=%-5s update %.2f ms wall (host enqueue %.2f ms)   graph replays so far %d, captures %d (%d in the timed stretch)" % (
            graph, prune, (t2 - t0) / ITERS * 1e3, (t1 - t0) / ITERS * 1e3, lr._fused.graph_replays, lr._fused.graph_captures,
            lr._fused.graph_captures - c0), flush=True)
if os.environ.get("PROFILE_HOST"):
    import cProfile
    import pstats

    Network.PRUNE_UNREACHABLE = True
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(20):
        lr.update()
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(55)
