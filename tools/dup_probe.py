#!/usr/bin/env python3
"""Which rows of a learner batch repeat?  (follow-up of obs_reuse_probe.py: 0.71 of the rows the online window encodes are duplicates
by value, the update's run-length reuse catches 0.19.)  Groups the needed rows by value and reports, per group, whether its members
belong to the same (window, agent), the same window, or different windows."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import mapf_rl_amd as M
from bench import heuristic_actions
from mapf_rl_amd.actor import VecActor
from mapf_rl_amd.learner import Learner
from mapf_rl_amd.replay import GlobalBuffer
from mapf_rl_amd.update import FusedUpdate
E, N, L = 1024, 40, 32
dev = torch.device("cuda"); torch.manual_seed(0)
lr = Learner(None, device=dev)
env = M.VecEnvironment(E, L, N, device=dev)
maps, agents, goals, _ = M.generate_scenarios(E, L, N, 0.3, seed=3); env.load(maps, agents, goals)
buf = GlobalBuffer(1 << (2 * E - 1).bit_length(), max_agents=N, device=dev, init_set=(N, L), fixed_level=True)
actor = VecActor(env, lr.model, buf, seed=0, density=0.3)
gen = torch.Generator(device=dev).manual_seed(5)
for k in range(300):
    actor.step(actions_override=heuristic_actions(actor.obs, gen).long())
learner = Learner(buf, device=dev, batch_size=192, model=lr.model)
fu = learner._fused
batch = buf.sample_batch(192)
FusedUpdate.DEDUP = True
pd = fu._finish_plan(fu.plan(batch))["online"]
tbp = pd.row_tbp[:pd.rows].long()
FusedUpdate.DEDUP = False
po = fu._finish_plan(fu.plan(batch))["online"]
FusedUpdate.DEDUP = True
R = po.rows
assert R == pd.rows
rows = po.obs_rows[:R].reshape(R, -1)
t, pos, b = tbp >> 24, (tbp >> 16) & 255, tbp & 0xFFFF
uniq, inv, cnt = torch.unique(rows.view(torch.int16), dim=0, return_inverse=True, return_counts=True)
print("rows %d, distinct by value %d (%.3f duplicates); run-length reuse keeps %d (%.3f dropped)" % (R, uniq.shape[0], 1 - uniq.shape[0] / R, pd.urows, 1 - pd.urows / R))
print("largest groups:", cnt.sort(descending=True).values[:10].tolist(), " rows that are all zero:", int(((rows != 0).sum(dim=1) == 0).sum()))
# distinct (value, window, position) and (value, window) combinations: what reuse restricted to one agent's window / one window could reach
key_bp = inv * (1 << 24) + b * 256 + pos
key_b = inv * (1 << 24) + b
print("distinct within (window, agent): %d (%.3f dropped)   distinct within a window: %d (%.3f dropped)" % (
    torch.unique(key_bp).numel(), 1 - torch.unique(key_bp).numel() / R, torch.unique(key_b).numel(), 1 - torch.unique(key_b).numel() / R))
# how the same-agent repeats are spread in time: gap to the previous occurrence of the same value for the same (window, agent)
order = torch.argsort(key_bp * 64 + t)
kb, tt = key_bp[order], t[order]
same = kb[1:] == kb[:-1]
gaps = (tt[1:] - tt[:-1])[same]
print("repeat gaps (steps) within (window, agent): 1: %d, 2: %d, 3-5: %d, >5: %d" % (int((gaps == 1).sum()), int((gaps == 2).sum()), int(((gaps >= 3) & (gaps <= 5)).sum()), int((gaps > 5).sum())))
