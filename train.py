"""Entry point compatible with the reference's `python3 train.py` (reference train.py:1-46): Ape-X style
training of the communicating DQN, with the 16 CPU actors + replay actor + GPU learner of the reference
replaced by one process per GPU that owns its vectorised environments, device replay and learner.

    python3 train.py                                   # 1 GPU, the reference's adaptive schedule
    python3 -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 train.py   # 8 GPUs, RCCL grad all-reduce
    python3 train.py --agents 40 --map 32 --envs 4096  # one fixed level (BASELINE config 2), no promotion

Schedule.  Like the reference (`Environment(adaptive=True)`, worker.py:362; config.init_set, pass_rate), the default
run starts at level (1 agent, 10x10), opens (agents+1, map) and (agents, map+5) when a level's last 200 episodes
reach 90 % success, and stops when every agent count has passed on the 40x40 map (worker.py:237-250).  `--envs` is
the number of lock-step environments PER ACTIVE LEVEL; the replay capacity defaults to what the active levels can
flush at once (see `--capacity`).  Passing `--agents` and/or `--map` trains ONE fixed level instead (a declared
deviation, for throughput runs): outcomes are still printed per level, but no level is promoted or retired.

Several ranks.  Every `learner.update()` contains the gradient all-reduce, so all ranks must run the same number of
updates: whether training has started (every rank's replay holds `--learning-starts` transitions), whether it is
time to print statistics / advance the curriculum (rank 0's clock), and whether to stop (`--minutes` on any rank,
or the curriculum's stop criterion on the POOLED level statistics) are decided every `--decide-every` iterations from
one small flag vector reduced asynchronously over a host-side (gloo) group one period earlier, identically on every
rank; no iteration waits for another rank outside the gradient exchange (reference worker.py:282-340: one learner
loop, no per-step barrier).

Printed statistics keep the reference's wording (worker.py:206-210,348-350): buffer update speed (= env steps/s
summed over this rank's environments), buffer size, per-level success, number of updates, update speed, loss.
"""
import argparse
import contextlib
import io
import os
import random
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this pool: RCCL needs it (see bench.py)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import config

torch.manual_seed(0)
np.random.seed(0)
random.seed(0)


def _pow2(n):
    """capacity * 256 leaves must be a power of two (buffer.py:23)."""
    return 1 << (int(n) - 1).bit_length()


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=512, help="lock-step environments per GPU (per active level with the curriculum).  The loop makes "
                    "one update per actor iteration and the curriculum is bound by updates (the stop criterion arrives ~100 k updates in, with the "
                    "reference's learning-rate milestone): 512 per level reach it in 207-231 s, 1024 in 257 s, 256 in 262 s (DESIGN section 7)")
    ap.add_argument("--agents", type=int, default=None, help="fixed level: number of agents (implies no curriculum)")
    ap.add_argument("--map", type=int, default=None, help="fixed level: map side length (implies no curriculum)")
    ap.add_argument("--curriculum", action="store_true", help="(default) the reference's adaptive schedule; kept for compatibility")
    ap.add_argument("--capacity", type=int, default=0, help="replay capacity in episodes; 0 = max(2048 (train.py:21), "
                    "2 x the episodes the active levels can flush in one iteration) so that the stale-priority filter of "
                    "update_priorities (worker.py:190-199) keeps working")
    ap.add_argument("--updates-per-iter", type=float, default=1.0, help="learner updates per actor iteration once training started")
    ap.add_argument("--max-updates", type=int, default=config.training_times)
    ap.add_argument("--minutes", type=float, default=0.0, help="stop after this many minutes (0 = until --max-updates / curriculum done); several ranks: "
                    "acted on up to 2 x --decide-every actor iterations late (the flag is reduced one decision period, read the next)")
    ap.add_argument("--interval", type=float, default=30.0, help="statistics interval in seconds (train.py:39); several ranks: same lag as --minutes")
    ap.add_argument("--promote-interval", type=float, default=5.0, help="seconds between two checks of the curriculum's promotion rule and stop "
                    "criterion (worker.py:211-224,237-250).  The reference checks them with the statistics, every 30 s of a loop that makes ~20 "
                    "updates/s; at this loop's ~400 updates/s the same wall-clock interval leaves a level that has long passed in place for "
                    "thousands of updates (to the stop criterion: 282 s with --interval 20, 232 s with --interval 10).  0 = with the "
                    "statistics only, as the reference does")
    ap.add_argument("--learning-starts", type=int, default=config.learning_starts, help="transitions in EVERY rank's replay before updates start "
                    "(several ranks: same lag as --minutes)")
    ap.add_argument("--batch-size", type=int, default=config.batch_size)
    ap.add_argument("--dist-backend", default="nccl", help="nccl = RCCL; gloo only for the CPU-side multi-rank tests")
    ap.add_argument("--double-q", action="store_true", default=bool(getattr(config, "double_q", False)),
                    help="double-DQN target (online argmax, target value); config.double_q is dead in the reference (worker.py:300-303)")
    ap.add_argument("--actor-update-steps", type=int, default=config.actor_update_steps,
                    help="actor iterations between two pulls of the learner's weights (config.actor_update_steps, worker.py:416-420)")
    ap.add_argument("--overlap-actors", type=int, default=-1, help="1: the actors' iteration runs on its own HIP stream beside the learner's "
                    "update (their episode flush ordered between two updates' replay operations by events); 0: one stream, strictly "
                    "alternating; -1 (default): 1 up to 48 agents per environment (curriculum, 5 minutes: 59 k -> 72.5 k updates; fixed 32x32 / "
                    "40-agent level: 63.6 -> 69.7 updates/s), 0 beyond (40x40 / 64 agents: 106.7 -> 61.3 updates/s, 64x64 / 128 agents: no change)")
    ap.add_argument("--decide-every", type=int, default=16, help="several ranks: actor iterations between two decision points (start of training, "
                    "statistics, stop); the flags are reduced asynchronously on host tensors and read one period later")
    ap.add_argument("--seed", type=int, default=0, help="initial weights, exploration and scenario streams")
    a = ap.parse_args(argv)
    if a.seed:  # (--seed 0 = the module-level seeds above: the runs of rounds 1-5)
        torch.manual_seed(a.seed)
        np.random.seed(a.seed)
        random.seed(a.seed)
    fixed = a.agents is not None or a.map is not None
    n_agents = a.agents if a.agents is not None else config.num_agents
    map_len = a.map if a.map is not None else config.map_length

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("MAPF_TRAIN_SHARE_GPU") == "1":
        local_rank = 0  # test mode: every rank on GPU 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(a.dist_backend)

    import mapf_rl_amd as M
    from mapf_rl_amd.actor import VecActor
    from mapf_rl_amd.learner import Learner
    from mapf_rl_amd.replay import GlobalBuffer

    seed = a.seed * 1000 + rank
    if fixed:
        capacity = a.capacity or _pow2(max(2048, 2 * a.envs))
        env = M.VecEnvironment(a.envs, map_len, n_agents, config.obs_radius, config.reward_fn, device=dev)
        maps, agents, goals, _ = M.generate_scenarios(a.envs, map_len, n_agents, -1.0, seed=seed)
        env.load(maps, agents, goals)
        buffer = GlobalBuffer(capacity, max_agents=max(n_agents, config.max_num_agetns), device=dev, init_set=(n_agents, map_len),
                              max_map_length=config.max_map_lenght, pass_rate=config.pass_rate, fixed_level=True)
    else:
        from mapf_rl_amd.curriculum import CurriculumActors

        # at most max_num_agetns + (max_map - init_map)/5 levels are active together (one per anti-diagonal step)
        max_levels = config.max_num_agetns + (config.max_map_lenght - config.init_set[1]) // 5
        capacity = a.capacity or _pow2(max(2048, 2 * a.envs * max_levels))
        buffer = GlobalBuffer(capacity, max_agents=config.max_num_agetns, device=dev, init_set=config.init_set,
                              max_map_length=config.max_map_lenght, pass_rate=config.pass_rate)
    learner = Learner(buffer, device=dev, batch_size=a.batch_size, save_path=config.save_path, double_q=a.double_q)
    if world > 1:  # identical initial weights on every rank
        if learner._fused is not None:
            # the parameters are views of ONE flat buffer: one collective; then the bf16 copy the kernels read follows explicitly (a write
            # through .data does not move the version counters FlatParams.sync() watches)
            dist.broadcast(learner._fused.flat.params, src=0)
            learner._fused.flat.refresh_bf16()
            learner.model.weights_epoch += 1
        else:
            for p in learner.model.parameters():
                dist.broadcast(p.data, src=0)
        learner.sync_target()
    # the actors act on a snapshot of the learner's weights, pulled every config.actor_update_steps iterations (worker.py:416-420)
    if fixed:
        actor = VecActor(env, learner.model, buffer, seed=seed, weights_period=a.actor_update_steps)
    else:
        actor = CurriculumActors(learner.model, buffer, envs_per_level=a.envs, device=dev, seed=seed, reward_fn=config.reward_fn,
                                 weights_period=a.actor_update_steps)

    # Actors beside the learner.  The reference's actors and learner are separate processes around a shared replay (worker.py); here
    # one process enqueues both, and with --overlap-actors the actor iteration goes to its own stream: at curriculum shapes an update
    # is a 4 ms chain of latency-bound launches and an actor iteration 1 ms of small ones, neither fills the chip.  The replay is the
    # only shared state: an actor iteration starts behind the previous update's replay operations (learner.replay_released) and the
    # next update's replay operations wait for it (learner.replay_gate); weight pulls wait for the learner's stream (actor.py).
    # (beyond 48 agents the actors' recurrence runs in the wide kernels, whole-CU workgroups for milliseconds: beside them the update's
    # chain of small launches starves -- measured, see --overlap-actors)
    overlap = ((n_agents if fixed else config.max_num_agetns) <= 48) if a.overlap_actors < 0 else bool(a.overlap_actors)
    from mapf_rl_amd.streams import role_stream

    astream = role_stream(dev, "actors") if overlap else None

    def actor_step():
        if astream is None:
            actor.step()
            return
        if learner.replay_released is not None:
            astream.wait_event(learner.replay_released)
        with torch.cuda.stream(astream):
            actor.step()
            done_ev = torch.cuda.Event()
            done_ev.record(astream)
        learner.replay_gate = done_ev

    # ---- decisions every rank must take identically (see module docstring) ----
    # Round 4 all-reduced three flags on the device and read them back EVERY iteration: a host-blocking collective per 0.3 ms actor
    # iteration.  Now the flags travel on a control-plane group of HOST tensors (gloo, whatever carries the gradients), once every
    # `--decide-every` iterations and asynchronously: the reduction issued at one decision point is read at the NEXT one, when it has
    # long completed, so the loop never waits for another rank outside the gradient exchange itself.  Every rank acts on the same
    # reduced flags at the same iteration number, hence identically (start of training, statistics, stop); the price is a lag of
    # one decision period.  The only blocking control-plane collective left is the pooled level statistics, once per interval.
    ctrl, K = None, 1
    if dist is not None:
        # (a group of its own whatever carries the gradients: with --dist-backend gloo the flags would otherwise share the WORLD group
        # with the host-staged gradient pieces, and correctness would hang on every rank issuing both in the same order)
        ctrl = dist.new_group(backend="gloo")
        K = max(1, a.decide_every)
    cpu = torch.device("cpu")
    pending = None
    t_start = t_last = t_promote = time.time()
    debt = 0.0
    started = False
    stop = False
    it = 0
    while learner.counter < a.max_updates and not stop:
        actor_step()
        it += 1
        stats_now = promote_now = 0
        if it % K == 0:
            if dist is None:
                now = time.time()
                if astream is not None and not started:
                    astream.synchronize()  # (nothing orders the actors' flush before the read below until updates run)
                not_ready = 0 if started else int(len(buffer) < a.learning_starts)  # (reads the device-side ring state)
                time_up = int(a.minutes > 0 and (now - t_start) > a.minutes * 60)
                stats_now = int(now - t_last >= a.interval)
                promote_now = int(not fixed and a.promote_interval > 0 and now - t_promote >= a.promote_interval)
            else:
                not_ready, time_up, stats_now, promote_now = int(not started), 0, 0, 0
                if pending is not None:
                    pending[0].wait()
                    not_ready, time_up, stats_now, promote_now = pending[1].tolist()
            if not started and not not_ready:
                started = True
                if rank == 0:
                    print("start training")
        if stats_now:
            now = time.time()
            if astream is not None:
                astream.synchronize()  # (the statistics below read what the actors wrote)
            pooled = buffer.pooled_counts(cpu, ctrl) if dist is not None else None
            with contextlib.nullcontext() if rank == 0 else contextlib.redirect_stdout(io.StringIO()):
                # per-rank buffers keep their own counters; only rank 0 prints (its own speed, the pooled level statistics)
                learner.stats(now - t_last)
                buffer.stats(now - t_last, pooled, world)
                print()
            if not fixed:
                actor.sync_levels()
                stop = buffer.check_done(pooled, world)  # worker.py:237-250, train.py:41-43
            if astream is not None:
                astream.wait_stream(torch.cuda.current_stream(dev))  # (new levels were set up on this stream)
            t_last = t_promote = now
        elif promote_now and not fixed:
            # the promotion rule and the stop criterion between two statistics (--promote-interval): same calls, nothing printed
            now = time.time()
            if astream is not None:
                astream.synchronize()
            pooled = buffer.pooled_counts(cpu, ctrl) if dist is not None else None
            buffer.advance_levels(pooled, world)
            actor.sync_levels()
            stop = buffer.check_done(pooled, world)
            if astream is not None:
                astream.wait_stream(torch.cuda.current_stream(dev))
            t_promote = now
        if it % K == 0:
            stop = stop or bool(time_up)
            if dist is not None and not stop:
                # this rank's flags as they are NOW (behind whatever the decisions above did), reduced by the next decision point
                now = time.time()
                if astream is not None and not started:
                    astream.synchronize()
                mine = torch.tensor([0 if started else int(len(buffer) < a.learning_starts),
                                     int(a.minutes > 0 and (now - t_start) > a.minutes * 60),
                                     int(rank == 0 and now - t_last >= a.interval),
                                     int(rank == 0 and not fixed and a.promote_interval > 0 and now - t_promote >= a.promote_interval)], dtype=torch.int32)
                pending = (dist.all_reduce(mine, op=dist.ReduceOp.MAX, group=ctrl, async_op=True), mine)
        if started:
            debt += a.updates_per_iter
            while debt >= 1.0 and learner.counter < a.max_updates:
                learner.update()
                debt -= 1.0
    if rank == 0:
        if stop and not fixed and not (a.minutes > 0 and (time.time() - t_start) > a.minutes * 60):
            print("stop criterion reached: {} updates, {:.0f} s after the start of the loop".format(learner.counter, time.time() - t_start))
        learner.save()
    if dist is not None:
        if pending is not None:
            pending[0].wait()  # (issued by every rank at the same decision point)
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
