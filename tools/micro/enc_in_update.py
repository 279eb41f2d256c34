#!/usr/bin/env python3
"""The encoder launches INSIDE a config-2 learner update against the same launches repeated back to back on the same buffers
(standalone the kernel takes 60 / 72 ns per observation at 12 k rows, without / with saved activations; inside an update the
timeline shows 85-97): HIP events around the two calls of an update, then each call replayed five times."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import mapf_rl_amd as M  # noqa: E402
from mapf_rl_amd import update as U  # noqa: E402
from mapf_rl_amd.actor import VecActor  # noqa: E402
from mapf_rl_amd.learner import Learner  # noqa: E402
from mapf_rl_amd.replay import GlobalBuffer  # noqa: E402
from bench import heuristic_actions  # noqa: E402

dev = torch.device("cuda")
N, L, E = 40, 32, 2048
torch.manual_seed(0)
buf = GlobalBuffer(4096, max_agents=N, device=dev, init_set=(N, L), fixed_level=True)
lr = Learner(buf, device=dev, batch_size=192)
env = M.VecEnvironment(E, L, N, device=dev)
maps, agents, goals, _ = M.generate_scenarios(E, L, N, 0.3, seed=1)
env.load(maps, agents, goals)
gen = torch.Generator(device=dev).manual_seed(11)
actor = VecActor(env, lr.model, buf, seed=0)
for _ in range(300):
    actor.step(actions_override=heuristic_actions(actor.obs, gen).long())
for _ in range(6):
    lr.update()
torch.cuda.synchronize()


class Spy:
    def __init__(self, real):
        self.real, self.calls = real, []

    def __getattr__(self, name):
        fn = getattr(self.real, name)
        if name not in ("mapf_encoder_forward", "mapf_encoder_forward_save"):
            return fn

        def wrapped(*args):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(torch.cuda.current_stream())
            rc = fn(*args)
            e1.record(torch.cuda.current_stream())
            self.calls.append((name, args, e0, e1))
            return rc

        return wrapped


spy = Spy(U.lib)
U.lib = spy
for it in range(3):
    spy.calls.clear()
    lr.update()
    torch.cuda.synchronize()
    calls = list(spy.calls)
    for name, args, e0, e1 in calls:
        inside = e0.elapsed_time(e1) * 1e3
        rows = args[2]
        fn = getattr(spy.real, name)
        a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fn(*args)
        a0.record()
        for _ in range(5):
            fn(*args)
        a1.record()
        torch.cuda.synchronize()
        alone = a0.elapsed_time(a1) / 5 * 1e3
        print("update %d  %-28s %6d rows: inside the update %7.1f us = %5.1f ns/obs   replayed back to back %7.1f us = %5.1f ns/obs" % (
            it, name, rows, inside, inside * 1e3 / rows, alone, alone * 1e3 / rows), flush=True)
