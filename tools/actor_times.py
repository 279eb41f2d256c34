#!/usr/bin/env python3
"""Steady-state actor iteration at the bench shape: wall / host-enqueue time with and without latent reuse, and the kernel list of one
iteration (torch.profiler).  Usage: actor_times.py [--tape]"""
import os, sys, time
import torch
from torch.profiler import ProfilerActivity, profile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import mapf_rl_amd as M
from bench import heuristic_actions
from mapf_rl_amd.actor import VecActor
from mapf_rl_amd.model import Network
from mapf_rl_amd.replay import GlobalBuffer
E, N, L = 4096, 40, 32
tape = "--tape" in sys.argv
dev = torch.device("cuda"); torch.manual_seed(0)
model = Network().cuda()
env = M.VecEnvironment(E, L, N, device=dev)
maps, agents, goals, _ = M.generate_scenarios(E, L, N, 0.3, seed=1); env.load(maps, agents, goals)
buf = GlobalBuffer(8192, max_agents=N, device=dev, init_set=(N, L), fixed_level=True)
actor = VecActor(env, model, buf, seed=0, density=0.3)
gen = torch.Generator(device=dev).manual_seed(3)
def step():
    actor.step(actions_override=heuristic_actions(actor.obs, gen).long() if tape else None)
for _ in range(300): step()
for reuse in (True, False):
    if not reuse: actor.latents = None
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): step()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    enc = actor.latents.last_encoded() / (E * N) if actor.latents is not None else 1.0
    print("reuse=%s tape=%s: %.2f ms per iteration (host enqueue %.2f ms), rows encoded %.3f" % (reuse, tape, (t2 - t0) / 20 * 1e3, (t1 - t0) / 20 * 1e3, enc), flush=True)
    if reuse:
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
            for _ in range(3): step()
            torch.cuda.synchronize()
        print(prof.key_averages().table(sort_by="device_time_total", row_limit=14, max_name_column_width=60))
if os.environ.get("PROFILE_HOST"):
    import cProfile, pstats
    actor.latents = actor.latents or None
    from mapf_rl_amd.fused import LatentCache
    if actor.latents is None:
        actor.latents = LatentCache()
    for _ in range(5): step()
    torch.cuda.synchronize()
    pr = cProfile.Profile(); pr.enable()
    for _ in range(40): step()
    torch.cuda.synchronize(); pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(40)
