#!/usr/bin/env python3
"""How much of a learner batch can influence the loss?  Only agent 0's Q-value at the window's last step is learned from
(reference model.py:248,255), and an agent's state reaches it only through the communication masks (two attention rounds per
step).  This probe fills the replay with the actor loop at config 2 and prints, for sampled batches, the fraction of (step, agent)
entries inside the backward closure of agent 0 -- for the online network's window (bt_steps) and the target's (bt_steps + steps)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mapf_rl_amd as M  # noqa: E402
from mapf_rl_amd.actor import VecActor  # noqa: E402
from mapf_rl_amd.model import Network, relevance  # noqa: E402
from mapf_rl_amd.replay import GlobalBuffer  # noqa: E402

E, L, N = int(os.environ.get("ENVS", 2048)), int(os.environ.get("MAP", 32)), int(os.environ.get("NAGENTS", 40))
dev = torch.device("cuda")
torch.manual_seed(0)
model = Network().to(dev)
env = M.VecEnvironment(E, L, N, device=dev)
maps, agents, goals, _ = M.generate_scenarios(E, L, N, 0.3, seed=1)
env.load(maps, agents, goals)
buf = GlobalBuffer(4096, max_agents=N, device=dev, init_set=(N, L), fixed_level=True)
from bench import heuristic_actions  # noqa: E402  (the executed actions while the replay fills: 80 % heuristic-following, so that the agents move)
hgen = torch.Generator(device=dev).manual_seed(11)
actor = VecActor(env, model, buf, seed=0)
for _ in range(int(os.environ.get("STEPS", 300))):
    actor.step(actions_override=heuristic_actions(actor.obs, hgen).long())
print("replay transitions:", len(buf))
for k in range(4):
    b = buf.sample_batch(192)
    comm, bt, steps = b[7], b[5], b[4].view(-1).long()
    for name, st in (("online", bt), ("target", bt + steps)):
        rel = relevance(comm, st)
        T = comm.shape[1]
        inside = (torch.arange(T, device=dev).view(T, 1) < st.view(1, -1)).unsqueeze(2).expand_as(rel)
        print("batch %d %-6s: %.3f of all (step, sample, agent) entries relevant; %.3f of those inside the windows; mean partners per row %.2f" % (
            k, name, float(rel.float().mean()), float(rel.float().sum() / inside.float().sum()), float(comm.float().sum(-1).mean())))
