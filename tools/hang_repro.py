#!/usr/bin/env python3
"""Two ranks sharing GPU 0 over gloo, each running learner updates with every observation encoded (the configuration in which
bench.py --gpus 2 hung intermittently in round 3).  Usage: hang_repro.py [--noside] [--prune] [--updates N]   (parent spawns 2 ranks)"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def rank_main():
    if os.environ.get("DBGLOG"):  # per-rank stderr file (AMD_LOG_LEVEL output)
        fd = os.open(os.path.join(ROOT, "gpurun_out", "hang_rank%s.err" % os.environ["RANK"]), os.O_WRONLY | os.O_CREAT | os.O_TRUNC)
        os.dup2(fd, 2)
    import faulthandler

    faulthandler.dump_traceback_later(int(os.environ.get("WATCHDOG", "45")), exit=True)
    import torch
    import torch.distributed as dist

    import mapf_rl_amd as M
    from bench import heuristic_actions
    from mapf_rl_amd.actor import VecActor
    from mapf_rl_amd.learner import Learner
    from mapf_rl_amd.model import Network
    from mapf_rl_amd.replay import GlobalBuffer

    rank = int(os.environ["RANK"])
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    E, L, N = 256, 32, 40
    torch.manual_seed(1234)
    maps, agents, goals, _ = M.generate_scenarios(E, L, N, 0.3, seed=1000 + rank)
    env = M.VecEnvironment(E, L, N, device=dev)
    env.load(maps, agents, goals)
    buf = GlobalBuffer(512, max_agents=N, device=dev, init_set=(N, L), fixed_level=True)
    lr = Learner(buf, device=dev, batch_size=192, prefetch="--noside" not in sys.argv)
    gen = torch.Generator(device=dev).manual_seed(7 + rank)
    actor = VecActor(env, lr.model, buf, seed=rank, density=0.3)
    for _ in range(260):
        actor.step(actions_override=heuristic_actions(actor.obs, gen).long())
    torch.cuda.synchronize()
    Network.PRUNE_UNREACHABLE = "--prune" in sys.argv
    n = int(sys.argv[sys.argv.index("--updates") + 1]) if "--updates" in sys.argv else 12
    t0 = time.time()
    for k in range(n):
        lr.update()
        if "--sync" in sys.argv:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    dist.barrier()
    print("rank %d: %d updates in %.2f s" % (rank, n, time.time() - t0), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    if "RANK" in os.environ:
        rank_main()
    else:
        import socket

        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:],
                                  env=dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))) for r in range(2)]
        rcs = [p.wait() for p in procs]
        print("exit codes", rcs, flush=True)
        sys.exit(max(rcs))
