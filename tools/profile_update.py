#!/usr/bin/env python3
"""Runs a few learner updates (and, with TACT, actor iterations) at the config-2 shape for rocprofv3 (kernel breakdown).  The
replay is filled by the actor loop itself, so the windows carry real communication masks (the share of observations that can
reach agent 0's Q-value -- and with it the encoder's work -- depends on them).  NAGENTS / MAPLEN / NENVS / TUPD / TACT / PRUNE."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import mapf_rl_amd as M
from mapf_rl_amd.actor import VecActor
from mapf_rl_amd.learner import Learner
from mapf_rl_amd.model import Network
from mapf_rl_amd.replay import GlobalBuffer
N = int(os.environ.get("NAGENTS", 40))
L = int(os.environ.get("MAPLEN", 64 if N > 64 else 32))
E = int(os.environ.get("NENVS", 1024 if N > 64 else 2048))
Network.PRUNE_UNREACHABLE = os.environ.get("PRUNE", "1") != "0"
if not Network.PRUNE_UNREACHABLE:  # "every observation": no pruning and no reuse of repeated observations either (the reference's work)
    from mapf_rl_amd.update import FusedUpdate
    FusedUpdate.DEDUP = False
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
torch.cuda.synchronize()
for _ in range(int(os.environ.get("TUPD", 4))):
    lr.update()
torch.cuda.synchronize()
if os.environ.get("TACT"):
    for _ in range(3):
        actor.step()
    torch.cuda.synchronize()
