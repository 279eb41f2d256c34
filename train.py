"""Entry point compatible with the reference's `python3 train.py` (reference train.py:1-46): Ape-X style
training of the communicating DQN, with the 16 CPU actors + replay actor + GPU learner of the reference
replaced by one process per GPU that owns its vectorised environments, device replay and learner.

    python3 train.py                                   # 1 GPU
    python3 -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 train.py   # 8 GPUs, RCCL grad all-reduce

Printed statistics keep the reference's wording (worker.py:206-210,348-350): buffer update speed (= env steps/s
summed over this rank's environments), buffer size, per-level success, number of updates, update speed, loss.
"""
import argparse
import os
import random
import time

import numpy as np
import torch

import config

torch.manual_seed(0)
np.random.seed(0)
random.seed(0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=1024, help="lock-step environments per GPU")
    ap.add_argument("--agents", type=int, default=config.num_agents)
    ap.add_argument("--map", type=int, default=config.map_length)
    ap.add_argument("--capacity", type=int, default=2048, help="replay capacity in episodes (train.py:21)")
    ap.add_argument("--updates-per-iter", type=float, default=1.0, help="learner updates per actor iteration once training started")
    ap.add_argument("--max-updates", type=int, default=config.training_times)
    ap.add_argument("--minutes", type=float, default=0.0, help="stop after this many minutes (0 = until --max-updates)")
    ap.add_argument("--interval", type=float, default=30.0, help="statistics interval in seconds (train.py:39)")
    ap.add_argument("--learning-starts", type=int, default=config.learning_starts)
    ap.add_argument("--batch-size", type=int, default=config.batch_size)
    ap.add_argument("--curriculum", action="store_true", help="reference schedule: start at config.init_set and promote levels "
                    "at config.pass_rate (worker.py:205-250); --envs is then the number of environments PER LEVEL")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    import mapf_rl_amd as M
    from mapf_rl_amd.actor import VecActor
    from mapf_rl_amd.learner import Learner
    from mapf_rl_amd.replay import GlobalBuffer

    if a.curriculum:
        from mapf_rl_amd.curriculum import CurriculumActors

        buffer = GlobalBuffer(a.capacity, max_agents=config.max_num_agetns, device=dev, init_set=config.init_set,
                              max_map_length=config.max_map_lenght, pass_rate=config.pass_rate)
    else:
        env = M.VecEnvironment(a.envs, a.map, a.agents, config.obs_radius, config.reward_fn, device=dev)
        maps, agents, goals, _ = M.generate_scenarios(a.envs, a.map, a.agents, -1.0, seed=rank)
        env.load(maps, agents, goals)
        buffer = GlobalBuffer(a.capacity, max_agents=max(a.agents, config.max_num_agetns), device=dev,
                              init_set=(a.agents, a.map), max_map_length=config.max_map_lenght, pass_rate=config.pass_rate)
    learner = Learner(buffer, device=dev, batch_size=a.batch_size, save_path=config.save_path)
    if world > 1:  # identical initial weights on every rank
        for p in learner.model.parameters():
            dist.broadcast(p.data, src=0)
        learner.sync_target()
    if a.curriculum:
        actor = CurriculumActors(learner.model, buffer, envs_per_level=a.envs, device=dev, seed=rank, reward_fn=config.reward_fn)
    else:
        actor = VecActor(env, learner.model, buffer, seed=rank)

    t_start = t_last = time.time()
    debt = 0.0
    started = False
    while learner.counter < a.max_updates:
        actor.step()
        if not started and len(buffer) >= a.learning_starts:
            started = True
            if rank == 0:
                print("start training")
        if started:
            debt += a.updates_per_iter
            while debt >= 1.0:
                learner.update()
                debt -= 1.0
        now = time.time()
        if now - t_last >= a.interval:
            if rank == 0:
                learner.stats(now - t_last)
                buffer.stats(now - t_last)
                print()
            else:  # per-rank buffers keep their own counters and levels; only rank 0 prints
                import contextlib
                import io

                with contextlib.redirect_stdout(io.StringIO()):
                    buffer.stats(now - t_last)
            if a.curriculum:
                actor.sync_levels()
                if buffer.check_done():  # worker.py:237-250, train.py:41-43
                    break
            t_last = now
        if a.minutes > 0 and (now - t_start) > a.minutes * 60:
            break
    if rank == 0:
        learner.save()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
