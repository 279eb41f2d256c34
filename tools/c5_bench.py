#!/usr/bin/env python3
"""BASELINE config 5 on one GPU (64x64 grid, 128 agents, 2048 environments; prioritized replay, optional double-DQN): env-step
kernel, full actor loop and learner update times.  The recurrence of 128-agent environments runs through
csrc/mapf_recur_wide*.hip.  Usage: c5_bench.py [agents] [map] [envs] [--double-q]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mapf_rl_amd as M  # noqa: E402
from mapf_rl_amd.actor import VecActor  # noqa: E402
from mapf_rl_amd.learner import Learner  # noqa: E402
from mapf_rl_amd.replay import GlobalBuffer  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
N = int(args[0]) if len(args) > 0 else 128
L = int(args[1]) if len(args) > 1 else 64
E = int(args[2]) if len(args) > 2 else 2048
dev = torch.device("cuda", 0)
torch.manual_seed(0)
env = M.VecEnvironment(E, L, N, device=dev)
env.reset_envs(None, 0.3, seed=1)
env.check_status()
from mapf_rl_amd.model import Network, relevance  # noqa: E402

# the replay is filled by the actor loop (random-init policy): the learner's windows then carry real communication masks, on which
# the share of observations that can reach agent 0's Q-value (model.relevance) -- the part an update encodes -- depends
cap = 1 << (2 * E - 1).bit_length()
buf = GlobalBuffer(cap, max_agents=N, device=dev, init_set=(N, L), fixed_level=True)
learner = Learner(buf, device=dev, batch_size=192, double_q="--double-q" in sys.argv)
from bench import heuristic_actions  # noqa: E402  (the executed actions while the replay fills: 80 % heuristic-following, so that the agents move)
hgen = torch.Generator(device=dev).manual_seed(11)
actor = VecActor(env, learner.model, buf, seed=0, density=0.3)
for _ in range(260):
    actor.step(actions_override=heuristic_actions(actor.obs, hgen).long())
torch.cuda.synchronize()
K = 12
t0 = time.perf_counter()
for _ in range(K):
    actor.step()
torch.cuda.synchronize()
dt_act = (time.perf_counter() - t0) / K


def timed_updates():
    for _ in range(5):  # (the row counts differ from batch to batch: let the caching allocator see a few)
        learner.update()
    torch.cuda.synchronize()
    t_ = time.perf_counter()
    for _ in range(K):
        learner.update()
    torch.cuda.synchronize()
    return (time.perf_counter() - t_) / K


dt_upd = timed_updates()
Network.PRUNE_UNREACHABLE = False
learner._drop_prefetch()
dt_all = timed_updates()
Network.PRUNE_UNREACHABLE = True
probe = buf.sample_batch(192)
reach = float(relevance(probe[7][:, :-2], probe[5]).float().mean())
env.check_status()
print("C5 shape %dx%d, %d agents, %d envs, double_q=%s: learner %.1f ms/update (%.1f updates/s, B=192 x T=18 x A=%d; %.3f of the window reachable; "
      "%.1f ms with every observation encoded), actor loop %.2f ms/iter (%.3g env-steps/s), peak memory %.1f GB" % (
          L, L, N, E, learner.double_q, dt_upd * 1e3, 1 / dt_upd, N, reach, dt_all * 1e3, dt_act * 1e3, E / dt_act,
          torch.cuda.max_memory_allocated() / 1e9), flush=True)
