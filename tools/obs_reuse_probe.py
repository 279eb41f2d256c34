#!/usr/bin/env python3
"""How many observation rows the encoder sees twice.  The encoder (reference model.py:147-162) is a deterministic per-observation
function, so an agent whose 6x9x9 observation did not change since the previous step needs no new encoder pass in the actor loop, and
identical rows inside a learner batch need one pass only -- exact reuse, not an approximation.  This probe measures, on the bench
configuration (32x32, 40 agents, rho = 0.3):
  actor    share of (environment, agent) rows per step whose bit-packed observation equals the previous step's, under
           (a) the tape policy (80 % heuristic-following / 20 % uniform), (b) the network's own greedy actions (random init, or a
           checkpoint given as --ckpt), averaged over the steps of whole episodes (auto-reset on);
  learner  share of duplicate rows among the rows a batch update encodes (online window).
Usage: obs_reuse_probe.py [--envs E] [--steps K] [--ckpt path.pth]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mapf_rl_amd as M  # noqa: E402
from bench import heuristic_actions  # noqa: E402
from mapf_rl_amd.actor import VecActor  # noqa: E402
from mapf_rl_amd.learner import Learner  # noqa: E402
from mapf_rl_amd.replay import GlobalBuffer  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=1024)
ap.add_argument("--steps", type=int, default=300)
ap.add_argument("--agents", type=int, default=40)
ap.add_argument("--map", type=int, default=32)
ap.add_argument("--ckpt", default=None)
args = ap.parse_args()
E, N, L = args.envs, args.agents, args.map
dev = torch.device("cuda")
torch.manual_seed(0)


def _unused_agent_rows(bits, N):
    """bit-packed observation rows [E, RD] int32 -> per-agent 486-bit rows as [E, N, 16] int32 words (61 bytes, zero padded)."""
    b = bits.view(torch.uint8).view(E, -1)
    idx = torch.arange(N * 486, device=dev).view(N, 486)
    bit = (b[:, (idx >> 3).view(-1)] >> (idx & 7).view(-1).to(torch.uint8)) & 1  # [E, N*486]
    return bit.view(E, N, 486)


def run(policy, label, model):
    env = M.VecEnvironment(E, L, N, device=dev)
    maps, agents, goals, _ = M.generate_scenarios(E, L, N, 0.3, seed=3)
    env.load(maps, agents, goals)
    buf = GlobalBuffer(1 << (2 * E - 1).bit_length(), max_agents=N, device=dev, init_set=(N, L), fixed_level=True)
    actor = VecActor(env, model, buf, seed=0, density=0.3)
    gen = torch.Generator(device=dev).manual_seed(5)
    prev = actor.obs.clone()
    same_tot, tot, by_phase = 0, 0, []
    for k in range(args.steps):
        if policy == "tape":
            fin = actor.step(actions_override=heuristic_actions(actor.obs, gen).long())
        else:
            fin = actor.step()
        cur = actor.obs
        same = (cur.view(E, N, -1) == prev.view(E, N, -1)).all(dim=2)      # [E, N]
        same = same & ~fin.bool().view(E, 1)                                # a reset environment starts over
        same_tot += int(same.sum())
        tot += E * N
        if k % 50 == 49 and actor.latents is not None:
            print("  step %d: LatentCache re-encoded %.3f of the rows at this step's start" % (k, actor.latents.last_encoded() / (E * N)))
        if k % 50 == 49:
            by_phase.append(float(same.float().mean()))
        prev = cur.clone()
    print("actor, %s: %.3f of the agent rows per step are identical to the previous step's (every 50th step: %s)" % (
        label, same_tot / tot, " ".join("%.2f" % v for v in by_phase)), flush=True)
    return buf


lr = Learner(None, device=dev)
if args.ckpt:
    lr.load_state_dict(torch.load(args.ckpt, map_location="cpu"))
buf = run("tape", "tape policy (80 % heuristic / 20 % uniform)", lr.model)
run("greedy", "the network's greedy actions (%s)" % ("checkpoint " + args.ckpt if args.ckpt else "random init"), lr.model)
# learner: duplicate rows among the rows an update encodes -- what the built reuse catches (same agent, consecutive steps: the plan's
# `urows` of `rows`) and the bound for any reuse scheme (distinct rows by value over the whole window set)
from mapf_rl_amd.update import FusedUpdate  # noqa: E402

learner = Learner(buf, device=dev, batch_size=192, model=lr.model)
fu = learner._fused
for _ in range(3):
    batch = buf.sample_batch(192)
    FusedUpdate.DEDUP = True
    po = fu._finish_plan(fu.plan(batch))["online"]
    built = 1 - po.urows / po.rows
    FusedUpdate.DEDUP = False
    po = fu._finish_plan(fu.plan(batch))["online"]
    FusedUpdate.DEDUP = True
    rows = po.obs_rows[:po.rows].reshape(po.rows, -1)
    torch.cuda.synchronize()
    uniq = torch.unique(rows.view(torch.int16), dim=0).shape[0]
    print("learner: %d rows to encode in the online window; the update's run-length reuse drops %.3f of them; %d distinct by value "
          "(%.3f duplicates: the bound)" % (po.rows, built, uniq, 1 - uniq / po.rows), flush=True)
