#!/usr/bin/env python3
"""Which Python-level operations the launches of ONE steady-state learner update come from: torch.profiler around a few updates at
the config-2 shape (replay filled by the actor loop under the tape policy), printed as (a) device time and launch count per
operator and (b) the in-order list of every device kernel of the last update with its duration and stream.
Usage: update_ops.py [agents] [map] [envs]   (env DOUBLE_Q=1, PRUNE=0)"""
import os
import sys
from collections import defaultdict

import torch
from torch.profiler import ProfilerActivity, profile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mapf_rl_amd as M  # noqa: E402
from mapf_rl_amd.actor import VecActor  # noqa: E402
from mapf_rl_amd.learner import Learner  # noqa: E402
from mapf_rl_amd.model import Network  # noqa: E402
from mapf_rl_amd.replay import GlobalBuffer  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
L = int(sys.argv[2]) if len(sys.argv) > 2 else 32
E = int(sys.argv[3]) if len(sys.argv) > 3 else 2048
Network.PRUNE_UNREACHABLE = os.environ.get("PRUNE", "1") != "0"
dev = torch.device("cuda")
torch.manual_seed(0)
buf = GlobalBuffer(4096, max_agents=max(N, 6), device=dev, init_set=(N, L), fixed_level=True)
lr = Learner(buf, device=dev, batch_size=192, double_q=os.environ.get("DOUBLE_Q") == "1")
env = M.VecEnvironment(E, L, N, device=dev)
maps, agents, goals, _ = M.generate_scenarios(E, L, N, 0.3, seed=1)
env.load(maps, agents, goals)
from bench import heuristic_actions  # noqa: E402

hgen = torch.Generator(device=dev).manual_seed(11)
actor = VecActor(env, lr.model, buf, seed=0)
for _ in range(300):
    actor.step(actions_override=heuristic_actions(actor.obs, hgen).long())
for _ in range(4):
    lr.update()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(3):
        lr.update()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="device_time_total", row_limit=70, max_name_column_width=70))
evs = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
evs.sort(key=lambda e: e.time_range.start)
marks = [i for i, e in enumerate(evs) if "encoder_bwd_kernel" in e.name]
if len(marks) >= 2:
    sel = evs[marks[-2]:marks[-1]]
    t0 = sel[0].time_range.start
    print("\none update between launches of encoder_bwd_kernel: %d device events, span %.2f ms" % (len(sel), (evs[marks[-1]].time_range.start - t0) / 1e3))
    by = defaultdict(lambda: [0.0, 0])
    for i, e in enumerate(sel):
        dur = e.time_range.end - e.time_range.start
        by[e.name[:60]][0] += dur
        by[e.name[:60]][1] += 1
        print("%4d  +%8.1f us  %8.1f us  %s" % (i, e.time_range.start - t0, dur, e.name[:120]))
    print("\nper kernel name:")
    for k, v in sorted(by.items(), key=lambda kv: -kv[1][0]):
        print("%9.1f us  x%-3d %s" % (v[0], v[1], k))
