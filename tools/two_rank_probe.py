#!/usr/bin/env python3
"""Where does a train-loop iteration go when several ranks run `bench.py`'s train loop?  (round-4 review: two ranks sharing one
GPU over gloo measured 410 ms per iteration beside a 14 ms update and a 4 ms actor iteration.)

Runs the bench's learner / actor / train-loop legs at the bench's shape with host-side stamps around every call that can block
(actor.step, learner.update, the gradient exchange's begin / finish) and prints one JSON line per rank.  Variants (`--variant`):

  real        the gradient exchange as the product issues it (gloo when several ranks share a GPU; nccl on one rank)
  none        begin / finish of the exchange replaced by no-ops (same ranks, same GPU sharing, no collective)
  serial      the real exchange, the actor iteration on the learner's stream (no second stream)

    python tools/two_rank_probe.py --ranks 2 --variant real        # starts its own ranks, all on GPU 0
    python tools/two_rank_probe.py --ranks 1 --variant real --force-dist nccl   # ONE rank with a 1-rank RCCL group forced into the >1-rank code path
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=2)
    ap.add_argument("--variant", default="real", choices=("real", "none", "serial"))
    ap.add_argument("--backend", default="gloo")
    ap.add_argument("--force-dist", default="", help="with --ranks 1: initialise a 1-rank process group of this backend and treat it as "
                    "several ranks (learner.FORCE_EXCHANGE): the >1-rank code path incl. the collective calls on one GPU")
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--agents", type=int, default=40)
    ap.add_argument("--map", type=int, default=32)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--updates", type=int, default=10)
    return ap.parse_args()


def launch(n, argv):
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK="0", WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env))
    rcs = [p.wait() for p in procs]
    return 1 if any(rcs) else 0


def main():
    a = parse()
    if "WORLD_SIZE" not in os.environ and a.ranks > 1:
        sys.exit(launch(a.ranks, sys.argv[1:]))
    import torch

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist = None
    if world > 1 or a.force_dist:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        be = a.force_dist or a.backend
        if be == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(be)
    import mapf_rl_amd as M
    from mapf_rl_amd import learner as learner_mod
    from mapf_rl_amd.actor import VecActor
    from mapf_rl_amd.learner import Learner
    from mapf_rl_amd.replay import GlobalBuffer

    if a.force_dist:
        learner_mod.FORCE_EXCHANGE = True
    sys.path.insert(0, ROOT)
    from bench import heuristic_actions

    E, L, N = a.envs, a.map, a.agents
    maps, agents, goals, _ = M.generate_scenarios(E, L, N, 0.3, seed=1000 + rank)
    env = M.VecEnvironment(E, L, N, device=dev)
    env.load(maps, agents, goals)
    gen = torch.Generator(device=dev)
    gen.manual_seed(77 + rank)
    torch.manual_seed(1234)
    cap = 1 << (2 * E - 1).bit_length()
    buf = GlobalBuffer(cap, max_agents=max(N, 6), device=dev, init_set=(N, L), fixed_level=True)
    learner = Learner(buf, device=dev, batch_size=192)
    actor = VecActor(env, learner.model, buf, seed=rank, density=0.3, weights_period=400)
    for _ in range(260):
        actor.step(actions_override=heuristic_actions(actor.obs, gen).long())
    torch.cuda.synchronize()

    out_iso = {}
    if dist is not None:
        # the collective in isolation (GPU otherwise idle): a host tensor of the gradient buffer's size, then the device buffer itself
        hostbuf = torch.ones(learner.bucket.flat.numel(), dtype=torch.float32)
        for name, t in (("host_tensor", hostbuf), ("device_tensor", learner.bucket.flat)):
            if name == "host_tensor" and dist.get_backend() == "nccl":  # (RCCL has no host path)
                continue
            ts = []
            for _ in range(6):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                dist.all_reduce(t)
                torch.cuda.synchronize()
                ts.append(round((time.perf_counter() - t0) * 1e3, 2))
            out_iso["allreduce_alone_ms_" + name] = ts
        learner.bucket.flat.zero_()
    stamps = {"begin": 0.0, "finish": 0.0, "n_begin": 0, "n_finish": 0}
    cls = type(learner.bucket)
    real_begin, real_finish = cls.begin, cls.finish
    if a.variant == "none":
        cls.begin = lambda self, lo, hi, group=None: None
        cls.finish = lambda self, group=None: None
    else:
        def begin(self, lo, hi, group=None):
            t = time.perf_counter()
            real_begin(self, lo, hi, group)
            stamps["begin"] += time.perf_counter() - t
            stamps["n_begin"] += 1

        def finish(self, group=None):
            t = time.perf_counter()
            real_finish(self, group)
            stamps["finish"] += time.perf_counter() - t
            stamps["n_finish"] += 1

        cls.begin, cls.finish = begin, finish

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    def reset_stamps():
        for k in stamps:
            stamps[k] = 0 if k.startswith("n_") else 0.0

    out = {"rank": rank, "world": world, "variant": a.variant, "backend": (dist.get_backend() if dist is not None else "none"),
           "graph_mode": bool(learner._fused is not None and learner._fused.graph_mode())}
    out.update(out_iso)
    # ---- learner alone ----
    for _ in range(5):
        learner.update()
    barrier()
    reset_stamps()
    t0 = time.perf_counter()
    host = 0.0
    for _ in range(a.updates):
        t = time.perf_counter()
        learner.update()
        host += time.perf_counter() - t
    torch.cuda.synchronize()
    out["learner_ms"] = (time.perf_counter() - t0) / a.updates * 1e3
    out["learner_host_ms"] = host / a.updates * 1e3
    out["learner_exchange_begin_ms"] = stamps["begin"] / a.updates * 1e3
    out["learner_exchange_finish_ms"] = stamps["finish"] / a.updates * 1e3
    barrier()
    # ---- actor alone (greedy) ----
    for _ in range(20):
        actor.step()
    barrier()
    t0 = time.perf_counter()
    host = 0.0
    for _ in range(a.iters):
        t = time.perf_counter()
        actor.step()
        host += time.perf_counter() - t
    torch.cuda.synchronize()
    out["actor_ms"] = (time.perf_counter() - t0) / a.iters * 1e3
    out["actor_host_ms"] = host / a.iters * 1e3
    barrier()
    # ---- the train loop ----
    astream = torch.cuda.Stream(device=dev) if a.variant != "serial" else None
    h = {"actor": 0.0, "update": 0.0}

    def train_iteration():
        t = time.perf_counter()
        if astream is None:
            actor.step()
        else:
            if learner.replay_released is not None:
                astream.wait_event(learner.replay_released)
            with torch.cuda.stream(astream):
                actor.step()
                ev = torch.cuda.Event()
                ev.record(astream)
            learner.replay_gate = ev
        t1 = time.perf_counter()
        learner.update()
        t2 = time.perf_counter()
        h["actor"] += t1 - t
        h["update"] += t2 - t1

    if astream is not None:
        astream.wait_stream(torch.cuda.current_stream(dev))
    train_iteration()
    barrier()
    reset_stamps()
    h["actor"] = h["update"] = 0.0
    per_iter = []
    t0 = time.perf_counter()
    for _ in range(a.iters):
        t = time.perf_counter()
        train_iteration()
        per_iter.append((time.perf_counter() - t) * 1e3)
    torch.cuda.synchronize()
    out["train_ms"] = (time.perf_counter() - t0) / a.iters * 1e3
    out["train_host_actor_ms"] = h["actor"] / a.iters * 1e3
    out["train_host_update_ms"] = h["update"] / a.iters * 1e3
    out["train_exchange_begin_ms"] = stamps["begin"] / a.iters * 1e3
    out["train_exchange_finish_ms"] = stamps["finish"] / a.iters * 1e3
    out["train_per_iter_ms"] = [round(v, 2) for v in per_iter]
    learner.replay_gate = None
    barrier()
    print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
