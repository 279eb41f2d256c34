#!/usr/bin/env python3
"""bench.py -- env steps/s of the vectorised MAPF environment hot path on MI355X.

Workload (BASELINE.json configs[1]): 32x32 grid, 40 agents, obstacle density 0.3, 4096 lock-step
environments per GPU.  One "step" = one pass of the fused step+observe kernel (mapf_step) over the
4096 resident environments, replaying a pre-recorded action tape (80 % heuristic-following / 20 %
uniform); inputs are resident in HBM before the timed region.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Multi-GPU: environments are independent, so rank g owns its own 4096 envs (weak scaling); there is no
data-path collective in the env step, only the timing barrier / max-over-ranks; the learner's gradient
all-reduce (RCCL) is exercised by the secondary `learner_updates_per_sec` rate.  Started WITHOUT a
launcher (`WORLD_SIZE` unset) and with `--gpus N > 1`, this process starts the N ranks itself -- before it
touches the GPU -- as fresh child processes (one per LOCAL_RANK, backend nccl = RCCL), forwards rank 0's
JSON line and exits non-zero if any rank failed.

Prints ONE JSON line on rank 0 (see the driver contract), including `roofline` for the dominant kernel
(env_step_kernel, HBM-bound), `cpu_baseline` (the CPU oracle timed on the host cores of this box) and the
system rates of the same pipeline (`learner_*`, `actor_loop_*`, `train_loop_*`, and -- the reference's OWN training
shape, B 192 x T 18 x <= 6 agents -- `learner_ref_shape_*`, `curriculum_actor_iter_*`) as top-level keys.

Several ranks: the secondary legs contain collectives (the learner's gradient exchange), so a rank that fails inside them
could leave the others waiting in an all-reduce until the driver's deadline -- and take the headline with it.  Hence, with
WORLD_SIZE > 1 only: rank 0 prints the headline line (metric, roofline, cpu_baseline; `"partial": true`) as soon as it exists and
the full superset line LAST; a rank that catches an exception in a secondary leg logs it and exits non-zero at once (the launcher
-- torch.distributed.run, or launch_ranks below -- then ends the other ranks within seconds), never walks on to a barrier; and
every rank carries a watchdog that dumps its stacks and exits after MAPF_BENCH_WATCHDOG seconds (default 480, below the driver's 600).
On one rank there is one line and a secondary-leg exception is reported as `dqn_error` beside the primary numbers.

Timed region: the K steps behind the W warm-up steps of the tape.  One launch is ~21 us, so K = 20 would be a 0.5 ms
sample: the K-step stretch is replayed R times back to back (one rewind launch of the agent positions between two
repetitions, inside the timed region and charged to it) until the region is >= ~12 ms; `steps` stays K, `timed_repeats` = R,
`ms_per_step` = elapsed / (K R).  `roofline.frac` is the BASELINE configuration's number (4096 environments: the 84 MB of navi
records + 80 MB of observations stay in the 256 MiB Infinity Cache between launches); `roofline.frac_out_of_cache` is the
same kernel on 4x the environments (observation writes go to HBM, the per-step navi reads still fit the cache) and
`roofline.frac_hbm_proper` on 8x (reads and writes both beyond the cache) -- see out_of_cache_leg.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

# the host driver of this pool supports dmabuf IPC only: RCCL (and CUDA-tensor sharing across processes) fails with
# hipIpcGetMemHandle: invalid argument without it; set before anything initialises the GPU (also when a launcher started this rank)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MIN_TIMED_S = 0.012   # the timed region is at least this long whatever --steps says (see the module docstring)
MALL_BYTES = 256 << 20  # Infinity Cache


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--envs", type=int, default=4096, help="environments per GPU")
    ap.add_argument("--map", type=int, default=32)
    ap.add_argument("--agents", type=int, default=40)
    ap.add_argument("--density", type=float, default=0.3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU-baseline wall time")
    ap.add_argument("--no-dqn", action="store_true", help="skip the secondary learner / actor-loop rates")
    ap.add_argument("--no-out-of-cache", action="store_true", help="skip the 4x-environments leg (roofline.frac_out_of_cache)")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL over xGMI) on real multi-GPU runs; gloo only to "
                    "exercise the multi-rank code path on a single GPU (set MAPF_BENCH_SHARE_GPU=1)")
    ap.add_argument("--dqn-updates", type=int, default=20)
    ap.add_argument("--dqn-actor-iters", type=int, default=12)
    ap.add_argument("--train-iters", type=int, default=10, help="interleaved actor-step + learner-update iterations")
    ap.add_argument("--no-ref-shape", action="store_true", help="skip the legs at the reference's own training shape (B 192 x T 18 x <= 6 agents)")
    ap.add_argument("--only-ref-shape", action="store_true", help="of the secondary legs, run only those at the reference's training shape (diagnostics)")
    ap.add_argument("--curriculum-envs", type=int, default=1024, help="environments per active curriculum level (train.py's default)")
    ap.add_argument("--curriculum-iters", type=int, default=200, help="timed curriculum actor iterations")
    ap.add_argument("--ref-shape-updates", type=int, default=100, help="timed updates at the reference's training shape (and train-loop pairs)")
    ap.add_argument("--ref-shape-warmup", type=int, default=150, help="untimed updates in front of them (graph captures of the buckets the replay produces)")
    return ap.parse_args(argv)


# --------------------------------------------------------------------------------------------------------
# rank launcher (parent process; never touches the GPU)
# --------------------------------------------------------------------------------------------------------
LAUNCH_DEADLINE_S = 540.0  # below the driver's 600 s: a job that hangs is ended HERE, with what rank 0 has printed forwarded


def launch_ranks(n, argv, script=None, deadline_s=None, grace_s=2.0):
    """Starts `n` ranks of this script as child processes and forwards rank 0's stdout.  The parent makes no HIP call
    (counting devices does not initialise the runtime on this image), so nothing GPU-initialised is ever re-executed.
    One rank exiting non-zero (or the deadline) ends all of them within `grace_s` + a poll; whatever rank 0 printed until then --
    the headline line -- is still forwarded.  `script` / `deadline_s`: tests (tests/test_bench_launcher_cpu.py)."""
    import torch

    share = os.environ.get("MAPF_BENCH_SHARE_GPU") == "1"
    have = torch.cuda.device_count()
    if have < n and not share:
        log("bench.py: --gpus %d requested but only %d HIP device(s) visible" % (n, have))
        return 2
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this pool (RCCL needs it)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(script or __file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, cwd=os.getcwd()))
    # rank 0's stdout is forwarded line by line AS IT ARRIVES (the headline line must not wait for the end of the job); the ranks are
    # polled so that one failing rank ends the others (they would otherwise wait for it at the next collective until the driver's timeout)
    import threading

    def forward():
        for line in iter(procs[0].stdout.readline, b""):
            sys.stdout.write(line.decode(errors="replace"))
            sys.stdout.flush()

    reader = threading.Thread(target=forward, daemon=True)
    reader.start()
    deadline = time.time() + (LAUNCH_DEADLINE_S if deadline_s is None else deadline_s)
    failed = False
    while any(p.poll() is None for p in procs):
        bad = [r for r, p in enumerate(procs) if p.poll() not in (None, 0)]
        if bad or time.time() > deadline:
            failed = True
            log("bench.py: %s -- ending the other ranks" % ("rank(s) %s exited non-zero" % bad if bad else "launcher deadline reached"))
            time.sleep(grace_s)  # let a rank that is already failing print its traceback
            for p in procs:
                if p.poll() is None:
                    p.kill()  # exactly the children started above
            break
        time.sleep(0.2)
    rcs = [p.wait() for p in procs]
    reader.join(timeout=10)
    if failed or any(rc != 0 for rc in rcs):
        log("bench.py: rank exit codes %s" % rcs)
        return 1
    return 0


def heuristic_actions(obs, gen, p_follow=0.8):
    """80 % follow a navi flag of the own cell (obs[:, :, 2:6, 4, 4]), 20 % uniform."""
    import torch

    E, N = obs.shape[:2]
    flags = obs[:, :, 2:6, 4, 4] != 0
    score = torch.rand((E, N, 4), device=obs.device, generator=gen) * flags
    follow = torch.where(flags.any(-1), 1 + score.argmax(-1), torch.zeros((), dtype=torch.long, device=obs.device))
    uni = torch.randint(0, 5, (E, N), device=obs.device, generator=gen)
    pick = torch.rand((E, N), device=obs.device, generator=gen) < p_follow
    return torch.where(pick, follow, uni).to(torch.int8).contiguous()


def heuristic_actions_rows(obs_rows, p_follow=0.8):
    """heuristic_actions for agent rows [R, 6, 9, 9] (the curriculum actors' shared observation buffer), drawing from the DEFAULT
    generator: capturable -- it runs inside the actors' replayed iteration (CurriculumActors.set_policy_override)."""
    import torch

    flags = obs_rows[:, 2:6, 4, 4] != 0
    R, d = flags.shape[0], obs_rows.device
    score = torch.rand((R, 4), device=d) * flags
    follow = torch.where(flags.any(-1), 1 + score.argmax(-1), torch.zeros((), dtype=torch.long, device=d))
    uni = torch.randint(0, 5, (R,), device=d)
    return torch.where(torch.rand((R,), device=d) < p_follow, follow, uni)


def _big_launch_point(M, dev, args, rank, E2, steps, warmup):
    """env_step_kernel on E2 environments of the bench's shape: own scenarios and tape, HIP events around `steps` back-to-back launches
    (second pass measured); the replay must reproduce the recording pass (positions) and its last observation block."""
    import torch

    L, N = args.map, args.agents
    maps, agents, goals, _ = M.generate_scenarios(E2, L, N, args.density, seed=5000 + rank)
    env = M.VecEnvironment(E2, L, N, device=dev)
    env.load(maps, agents, goals)
    gen = torch.Generator(device=dev)
    gen.manual_seed(177 + rank)
    T = steps + warmup
    tape = torch.empty((T, E2, N), dtype=torch.int8, device=dev)
    obs, pos = env.observe()
    for t in range(T):
        tape[t] = heuristic_actions(obs, gen)
        obs, pos, *_ = env.step(tape[t])
    first = pos.clone()
    first_obs = obs.clone()
    a0 = torch.from_numpy(agents).to(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for rep in range(2):  # the second pass is the measured one
        env.set_agents(a0)
        for t in range(warmup):
            env.step(tape[t])
        e0.record()
        for t in range(warmup, T):
            obs, *_ = env.step(tape[t])
        e1.record()
    torch.cuda.synchronize()
    env.check_status()
    assert torch.equal(env.pos, first), "large-launch replay diverged from its recording pass"
    assert torch.equal(obs, first_obs), "large-launch replay: observation bytes differ from the recording pass"
    us = e0.elapsed_time(e1) * 1e3 / steps
    del env, tape, first_obs
    return us


def out_of_cache_leg(M, dev, args, rank, steps=40, warmup=8):
    """The same kernel beyond the 256 MiB Infinity Cache (MALL), at two sizes, because "out of cache" has two thresholds:

    * `*_out_of_cache` -- 4x the environments (config 2: 16,384).  The observation bytes one launch WRITES (318 MB) exceed the MALL and
      go to HBM (non-temporal stores, step_nt_store); the navi records one step READS -- about 10 of an agent's 32 rows, ~180 MB by
      FETCH_SIZE -- still fit it, so this point is HBM writes + largely MALL-served reads.  (Rounds 2-3 called it "nothing stays
      cached": wrong for the reads.)
    * `*_hbm_proper` -- the smallest power-of-two multiple whose per-step READ set also exceeds the MALL (config 2: 32,768
      environments, ~360 MB read + 637 MB written per launch): both directions stream from / to HBM.  This is the fraction to hold
      against the 8 TB/s spec; the chip's plain-copy rate is 6.29 TB/s = 0.79 (MI355X_MICROARCH.md).
    * `*_hbm_proper_2x` -- twice that again (65,536): the read set is 3x the MALL; whatever residual hits the 32,768 point still had
      are gone (measured 0.84 -> 0.79).
    `read_set_bytes_*` = what one step fetches, estimated per environment as map rows + N x (10 navi records at 128-byte line
    granularity) + positions / goals / actions; the measured FETCH_SIZE of the points is in profiles/r04_shape_sweep.md."""
    L, N = args.map, args.agents
    wb = 4 if L <= 32 else 8
    per_env = N * L * 4 * wb + N * 486  # resident navi records + observation bytes
    # one step's fetches at cache-line granularity: an agent's ~10 consecutive navi records (16 / 32 B each) straddle 128-byte lines
    # (config 2: 12.0 KB per environment against 11.0 KB by FETCH_SIZE at 16,384 environments, profiles/r04_shape_sweep.md)
    read_env = L * wb + N * (10 * 4 * wb + 128) + 9 * N
    alg_env = L * L + 821 * N + 1
    E_w = 4 * args.envs
    while E_w * per_env < 2 * MALL_BYTES:
        E_w *= 2
    E_h = E_w
    while E_h * read_env < 1.25 * MALL_BYTES:
        E_h *= 2
    out = {}
    for tag, E2 in (("out_of_cache", E_w), ("hbm_proper", E_h), ("hbm_proper_2x", 2 * E_h)):
        us = _big_launch_point(M, dev, args, rank, E2, steps, warmup)
        ach = alg_env * E2 / (us * 1e-6) / 1e9
        out.update({"envs_" + tag: E2, "working_set_%s_bytes" % tag: E2 * per_env, "read_set_bytes_" + tag: E2 * read_env,
                    "kernel_avg_us_" + tag: us, "achieved_" + tag: ach, "frac_" + tag: ach / HBM_PEAK_GBS})
    out["out_of_cache_note"] = ("frac_out_of_cache: observation writes exceed the 256 MiB Infinity Cache, the per-step navi reads do not; "
                                "frac_hbm_proper: reads and writes both exceed it; frac_hbm_proper_2x: twice the environments again")
    return out


NOMINAL_GHZ = 2.4  # the clock the 2.5 PFLOP/s dense bf16 / f16 peak is quoted at (MI355X_MICROARCH.md)


def encoder_clock_ghz(model, obs_flat, dev, seconds=2.0):
    """The shader clock the chip HOLDS under the encoder kernel: delta s_memtime / delta s_memrealtime x 100 MHz, stamped once per
    workgroup in the diagnostic build of the same kernel (mapf_rl_amd/libmapf_enc_clock.so: csrc/mapf_encoder.hip with -DMAPF_ENC_CLOCK;
    in the product's kernel no stamp executes), after `seconds` of back-to-back launches on the bench's observations, median over
    workgroups (MI355X_MICROARCH.md, DVFS give-back item 6).  Returns the keys it adds to `encoder_roofline`; {} without the library."""
    import ctypes

    import numpy as np
    import torch

    path = os.path.join(ROOT, "mapf_rl_amd", "libmapf_enc_clock.so")
    if not os.path.exists(path):
        return {"clock_note": "libmapf_enc_clock.so not built"}
    try:
        lib = ctypes.CDLL(path)
        fwd, rd = lib.mapf_encoder_forward, lib.mapf_enc_clock_read
        fwd.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        rd.argtypes = [ctypes.c_void_p, ctypes.c_int]
        wp, bp = model._packed.get(model.obs_encoder, model.weights_epoch)
        obs = obs_flat.contiguous()
        assert obs.dtype == torch.uint8
        M = obs.shape[0]
        lat = torch.empty((M, 784), dtype=torch.bfloat16, device=dev)
        st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        args = (ctypes.c_void_p(obs.data_ptr()), 0, M,  # (0 = MAPF_ENC_OBS_U8)
                ctypes.c_void_p(wp.data_ptr()), ctypes.c_void_p(bp.data_ptr()), ctypes.c_void_p(lat.data_ptr()), st)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        assert fwd(*args) == 0
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 0
        while time.perf_counter() - t0 < seconds:  # back to back: the queue never runs dry (20 launches of ~10 ms per check)
            for _ in range(20):
                fwd(*args)
            n += 20
            if n % 100 == 0:  # (bounds the queue: ~1 s of launches ahead at most)
                torch.cuda.current_stream(dev).synchronize()
        e0.record()
        for _ in range(5):
            fwd(*args)
        e1.record()
        torch.cuda.synchronize()
        nwg = min(8192, (M + 3) // 4)
        buf = np.zeros((nwg, 2), dtype=np.uint64)
        assert rd(buf.ctypes.data_as(ctypes.c_void_p), nwg) == 0
        ok = buf[:, 1] > 0
        ghz = float(np.median(buf[ok, 0].astype(np.float64) / buf[ok, 1].astype(np.float64))) * 0.1
        ms = e0.elapsed_time(e1) / 5
        flop = 2.0 * (49 * 128 * 54 + 6 * 49 * 128 * 1152 + 49 * 16 * 128) * M
        ach = flop / (ms * 1e-3) / 1e12
        return {"clock_ghz": ghz, "peak_at_clock": 2500.0 * ghz / NOMINAL_GHZ, "frac_at_clock": ach / (2500.0 * ghz / NOMINAL_GHZ),
                "clock_build_kernel_avg_ms": ms, "clock_build_achieved": ach,
                "clock_note": "median over %d workgroups of delta s_memtime / delta s_memrealtime x 100 MHz in the diagnostic build of the same kernel "
                              "after %.1f s of back-to-back launches; frac_at_clock = that build's TFLOP/s over 2500 x clock / %.1f GHz" % (int(ok.sum()), seconds, NOMINAL_GHZ)}
    except Exception as ex:  # the primary numbers stand
        return {"clock_note": "clock measurement failed: %r" % (ex,)}


def cpu_baseline(args, maps, agents, goals, tape, final_pos, E, T):
    """The oracle (C restatement with the reference's sequential semantics), in child processes without torch and without
    the GPU: one single-threaded, pinned process per usable host core (oracle/cpu_bench.py)."""
    import tempfile

    import numpy as np

    ncpu = len(os.sched_getaffinity(0))
    S = min(E, max(256, 4 * ncpu))
    with tempfile.TemporaryDirectory() as td:
        f = os.path.join(td, "sample.npz")
        np.savez(f, maps=maps[:S], agents=agents[:S], goals=goals[:S], tape=tape[:, :S].cpu().numpy())
        out = subprocess.run([sys.executable, "-m", "oracle.cpu_bench", f, str(args.cpu_seconds)],
                             cwd=ROOT, capture_output=True, text=True, timeout=max(120.0, 6 * args.cpu_seconds))
    assert out.returncode == 0, out.stderr[-2000:]
    cb = json.loads(out.stdout.strip().splitlines()[-1])
    # trajectories must be identical to the GPU's before any number is reported
    assert np.array_equal(np.array(cb["final_agents"], np.int16), final_pos[:S].cpu().numpy()), "CPU/GPU trajectories differ"
    # `cores` = the host cores that actually did the work: the worker processes, capped by the cgroup CPU quota when the box has one
    # (32 pinned workers on a 16-core quota time-slice 16 cores); `workers` = the processes the scan picked
    quota = cb.get("cpu_quota_cores")
    cores = cb["workers"] if not quota else min(cb["workers"], max(1, int(quota + 0.5)))
    return {
        "value": cb["env_steps_per_sec"], "unit": "env-steps/s", "cores": cores, "workers": cb["workers"], "kind": "port",
        "sample": "first %d envs x %d tape steps, step+observe, repeated for %.1f s by %d single-threaded oracle processes, each pinned "
                  "to one CPU (%s); trajectories verified identical to the GPU run" % (S, T, cb["seconds"], cb["workers"], cb["how"]),
        "cpu_model": cb["cpu_model"], "logical_cpus": cb["logical_cpus"], "physical_cores": cb["physical_cores"],
        "cpu_quota_cores": cb["cpu_quota_cores"], "worker_scan": cb["scan"],
    }


# --------------------------------------------------------------------------------------------------------
# several ranks: fail fast, never hang
# --------------------------------------------------------------------------------------------------------
WATCHDOG_DEFAULT_S = 480  # several ranks: every rank dumps its stacks and exits after this long (the driver's deadline is 600 s)


def fault_point(stage, rank):
    """Tests / rehearsals: MAPF_BENCH_FAULT="<rank>:<stage>[:hang]" makes that rank raise (or stop responding) at that stage."""
    spec = os.environ.get("MAPF_BENCH_FAULT")
    if not spec:
        return
    f = spec.split(":")
    if int(f[0]) == rank and f[1] == stage:
        if len(f) > 2 and f[2] == "hang":
            log("[rank %d] injected hang at stage %r" % (rank, stage))
            time.sleep(10 ** 6)
        raise RuntimeError("injected fault at stage %r on rank %d (MAPF_BENCH_FAULT)" % (stage, rank))


def abort_job(stage, rank, ex):
    """A rank of a multi-rank job that caught an exception where the other ranks may be inside a collective: say so and exit NOW,
    non-zero, without running destructors that could wait for the GPU or the process group (the launcher ends the other ranks)."""
    import traceback

    log("[rank %d] bench.py: %s failed: %r -- leaving the job (exit 13) so that no rank waits in a collective" % (rank, stage, ex))
    traceback.print_exc()
    sys.stderr.flush()
    sys.stdout.flush()
    os._exit(13)


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    wd = os.environ.get("MAPF_BENCH_WATCHDOG") or (str(WATCHDOG_DEFAULT_S) if world > 1 else "")
    if wd and int(wd) > 0:  # dump every thread's stack and exit (non-zero) if the run takes longer than this
        import faulthandler

        faulthandler.dump_traceback_later(int(wd), exit=True)

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if os.environ.get("MAPF_BENCH_SHARE_GPU") == "1":
        local_rank = 0  # test mode: every rank on GPU 0
    assert torch.cuda.is_available(), "bench.py needs a HIP device"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    import mapf_rl_amd as M

    E, L, N, K, W = args.envs, args.map, args.agents, args.steps, args.warmup
    T = K + W
    t0 = time.time()
    maps, agents, goals, redraws = M.generate_scenarios(E, L, N, args.density, seed=1000 + rank)
    env = M.VecEnvironment(E, L, N, device=dev)
    env.load(maps, agents, goals)
    torch.cuda.synchronize()
    log("[rank %d] scenarios + navi ready in %.2fs (redraws %d)" % (rank, time.time() - t0, redraws))

    # ---- record the action tape by running the policy once (untimed) ----
    gen = torch.Generator(device=dev)
    gen.manual_seed(77 + rank)
    tape = torch.empty((T, E, N), dtype=torch.int8, device=dev)
    obs, pos = env.observe()
    for t in range(T):
        tape[t] = heuristic_actions(obs, gen)
        obs, pos, rew, done, rc = env.step(tape[t])
    env.check_status()
    final_pos_first_pass = pos.clone()
    agents_dev = torch.from_numpy(agents).to(dev)

    # ---- untimed replays ----
    # (two untimed passes over the whole tape first: the recording pass above ran interleaved with the policy's own kernels, and a
    # profile of this command should be dominated by launches in the regime the timed region measures)
    e_a, e_b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for rep in range(2):
        env.set_agents(agents_dev)
        if rep:
            e_a.record()
        for t in range(T):
            env.step(tape[t])
    e_b.record()
    torch.cuda.synchronize()
    est_step_s = max(e_a.elapsed_time(e_b) * 1e-3 / T, 1e-6)
    R = int(min(2000, max(1, -(-MIN_TIMED_S // (K * est_step_s)))))  # repetitions of the K-step stretch: timed region >= ~12 ms

    # ---- CPU baseline (rank 0) BEFORE the process group exists: the other ranks then sleep in the rendezvous instead of
    # spinning on their GPUs / holding host cores while the oracle processes are being timed ----
    cpu = cpu_err = None
    if rank == 0 and not args.no_cpu_baseline:
        try:
            cpu = cpu_baseline(args, maps, agents, goals, tape, final_pos_first_pass, E, T)
        except Exception as ex:
            # reported in the line (`cpu_baseline_error`), never fatal: on several ranks the others are waiting in the rendezvous for
            # this rank, and a job that dies here prints nothing at all
            import traceback

            traceback.print_exc()
            cpu_err = repr(ex)[:400]
    ranks_seen = None
    if world > 1:
        import datetime

        # (the timeout bounds the rendezvous -- ranks 1.. wait here while rank 0 times the CPU baseline -- and, with RCCL, every
        # collective: a rank that never arrives ends the others through the backend's own watchdog well inside the driver's deadline)
        tmo = datetime.timedelta(seconds=int(os.environ.get("MAPF_BENCH_PG_TIMEOUT", "300")))
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, timeout=tmo)
        else:
            dist.init_process_group(args.dist_backend, timeout=tmo)
        assert dist.get_world_size() == world
        rr = torch.tensor([R], dtype=torch.int64, device=dev)
        dist.all_reduce(rr, op=dist.ReduceOp.MAX)  # same amount of timed work on every rank
        R = int(rr.item())
        # who is here: (rank, HIP device index, PCI bus id) of every rank, gathered once
        me = torch.zeros((world, 3), dtype=torch.int64, device=dev)
        try:
            bus = int(str(torch.cuda.get_device_properties(dev).pci_bus_id))
        except Exception:
            bus = -1
        me[rank] = torch.tensor([rank, local_rank, bus], dtype=torch.int64, device=dev)
        dist.all_reduce(me, op=dist.ReduceOp.SUM)
        ranks_seen = [{"rank": int(a), "device": int(b), "pci_bus": int(c)} for a, b, c in me.tolist()]

    # ---- timed replay ----
    env.set_agents(agents_dev)
    for t in range(W):
        env.step(tape[t])
    pos_w = env.agents_pos()  # state behind the warm-up steps: the rewind target of repetitions 2..R
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(R)]
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t_start = time.perf_counter()
    for r in range(R):
        if r:
            env.set_agents(pos_w, sync=False)  # one small launch, inside the timed region
        evs[r][0].record()  # HIP events on the launch stream around each K-launch stretch (no per-launch event traffic)
        for k in range(K):
            env.step(tape[W + k])
        evs[r][1].record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t_start
    env.check_status()
    assert torch.equal(env.pos, final_pos_first_pass), "replay diverged from the recording pass"

    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    kern_avg_s = sum(a.elapsed_time(b) for a, b in evs) * 1e-3 / (K * R)  # mean launch-to-launch duration of env_step_kernel (incl. the ~1 us boundary)
    alg_bytes_per_env = L * L + 821 * N + 1  # fused step+observe, one byte per flag/cell
    alg_bytes = alg_bytes_per_env * E
    achieved = alg_bytes / kern_avg_s / 1e9

    backend = "none" if world == 1 else ("%s, dist.get_world_size()=%d" % (dist.get_backend(), dist.get_world_size()))
    result = {
        "metric": "env_steps_per_sec",
        "value": world * E * K * R / elapsed,
        "unit": "env-steps/s",
        "n_gpus": world,
        "steps": K,
        "warmup": W,
        "ms_per_step": elapsed / (K * R) * 1e3,
        "timed_repeats": R,
        "timed_region_ms": elapsed * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u8",
        "data": "synthetic",
        "config": {"workload": "mapf env step+observe, %dx%d grid, %d agents, rho=%.2f, %d envs/GPU" % (L, L, N, args.density, E),
                   "map": L, "agents": N, "envs_per_gpu": E, "obs_radius": 4,
                   "parallelism": "env-sharded x%d (%s)" % (world, backend)},
        "roofline": {"bound": "hbm", "kernel": "env_step_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None, "traffic_source": None,
                     "alg_bytes_per_launch": alg_bytes, "kernel_avg_us": kern_avg_s * 1e6},
    }
    tr = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tr) and (E, L, N) == (4096, 32, 40):  # the committed PMC passes were taken on this workload
        try:
            tj = json.load(open(tr))
            result["roofline"]["traffic"] = tj.get("env_step_kernel_bytes_per_launch")
            # NOT measured in this run: PMC counters need rocprofv3 around the process (separate --pmc passes)
            result["roofline"]["traffic_source"] = "profiles/traffic.json (%s)" % tj.get("source", "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes")
            # the same counters at the two larger launches (same kernel, same passes): per-launch HBM bytes beside frac_out_of_cache / frac_hbm_proper
            result["roofline"]["traffic_out_of_cache"] = tj.get("e16384_bytes_per_launch")
            result["roofline"]["traffic_hbm_proper"] = tj.get("e32768_bytes_per_launch")
        except Exception:
            pass

    if cpu is not None:
        result["cpu_baseline"] = cpu
    if cpu_err is not None:
        result["cpu_baseline_error"] = cpu_err
    if world > 1:
        result["multi_rank"] = multi_rank_info(torch, dist, ranks_seen, args)
        if rank == 0:
            # the headline exists: on stdout NOW, before any leg that contains a collective (module docstring); the full line follows last
            print(json.dumps(dict(result, partial=True, partial_note="headline only: printed before the secondary legs; the last JSON line "
                                                                    "of this run is the complete one")), flush=True)
    fault_point("after_headline", rank)
    if not args.no_out_of_cache:
        try:  # (no collective inside: a rank that fails here reports it and goes on with the others)
            result["roofline"].update(out_of_cache_leg(M, dev, args, rank))
        except Exception as ex:  # (e.g. not enough free memory next to another process): the primary numbers stand
            result["roofline"]["out_of_cache_error"] = repr(ex)[:200]

    # ---- system rates of the same pipeline (BASELINE metric: "env steps/sec + learner updates/sec") ----
    if not args.no_dqn:
        for name, leg in (("dqn legs (BASELINE configs[1] shape)", dqn_legs), ("reference-training-shape legs", ref_shape_legs)):
            if (leg is ref_shape_legs and args.no_ref_shape) or (leg is dqn_legs and args.only_ref_shape):
                continue
            try:
                result.update(leg(M, args, env, dev, rank, world, dist, gen))
            except Exception as ex:
                if world > 1:  # the other ranks may be inside a gradient all-reduce: leave, do not walk on to a barrier
                    abort_job(name, rank, ex)
                import traceback

                traceback.print_exc()
                result.setdefault("dqn_error", "")
                result["dqn_error"] += "%s: %s; " % (name, repr(ex)[:300])  # the primary metric must still be reported
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def multi_rank_info(torch, dist, ranks_seen, args):
    """What a per-N comparison needs to know about the job: backend + library version, who took part, the knobs in force."""
    info = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "ranks_seen": ranks_seen,
            "env": {k: v for k, v in sorted(os.environ.items()) if k.split("_")[0] in ("NCCL", "RCCL", "HSA", "HIP", "GPU", "ROCR", "TORCH")
                    and "KEY" not in k and "TOKEN" not in k}}
    try:
        info["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception as ex:
        info["rccl_version"] = "unknown (%r)" % (ex,)
    info["shared_gpu_rehearsal"] = os.environ.get("MAPF_BENCH_SHARE_GPU") == "1"
    return info


def exchange_stats(bucket, per):
    """Mean per-update cost of the gradient exchange from learner.FlatGradBucket's timing records (`per` updates): host time inside
    begin() / finish(), and the time the update's stream spent between the start of finish() and the averaged gradients being ready
    (HIP events on that stream: the EXPOSED part of the exchange -- what overlapped with the backward chain does not show)."""
    rec = bucket.timing or []
    if not rec or per <= 0:
        return {}
    b = [r[1] for r in rec if r[0] == "begin"]
    f = [r for r in rec if r[0] == "finish"]
    out = {"exchange_begin_ms": sum(b) * 1e3 / per, "exchange_finish_host_ms": sum(r[1] for r in f) * 1e3 / per,
           "exchange_pieces_per_update": len(b) / per}
    ev = [r[2].elapsed_time(r[3]) for r in f if r[2] is not None]
    if ev:
        out["exchange_finish_ms"] = sum(ev) / per
    return out


def dqn_legs(M, args, env, dev, rank, world, dist, gen):
    """learner: Learner.update on 192 x 18 x 40 windows sampled from the device replay (bf16, incl. the flat gradient all-reduce over
    RCCL when world > 1); actor loop: policy inference + env step + recording; train loop: one actor iteration + one learner
    update per iteration, the actor on its own stream beside the update (train.py --overlap-actors)."""
    import torch

    from mapf_rl_amd.actor import VecActor
    from mapf_rl_amd.learner import Learner
    from mapf_rl_amd.model import Network, relevance
    from mapf_rl_amd.replay import GlobalBuffer
    from mapf_rl_amd.update import FusedUpdate

    E, N = args.envs, args.agents
    torch.manual_seed(1234)  # identical initial weights on every rank
    # The replay is filled by the actor loop itself on the bench's scenarios: the learner's windows then carry REAL
    # communication masks.  That matters: only agent 0's Q-value is learned from, so the update encodes just the
    # observations that can reach it through the masks (model.relevance) -- with random mask bits everything would be
    # reachable.  The agents MOVE while the replay fills: the executed actions are the tape policy's (80 % heuristic-
    # following; the random-init network's own greedy actions leave most agents standing, which would understate the
    # reachable share: 0.12 instead of 0.20).  Episodes enter the replay when they end or time out (256 steps), hence the
    # 260 iterations.
    cap = 1 << (2 * E - 1).bit_length()
    buf = GlobalBuffer(cap, max_agents=max(N, 6), device=dev, init_set=(N, args.map), fixed_level=True)
    learner = Learner(buf, device=dev, batch_size=192)
    import config as ref_config

    # the actor acts on a snapshot of the learner's weights pulled every config.actor_update_steps = 400 iterations, as the
    # reference's actors do (worker.py:416-420) and as train.py runs it
    actor = VecActor(env, learner.model, buf, seed=rank, density=args.density, weights_period=ref_config.actor_update_steps)
    tape_actions = lambda: heuristic_actions(actor.obs, gen).long()
    for _ in range(260):
        actor.step(actions_override=tape_actions())
    torch.cuda.synchronize()
    fault_point("dqn", rank)

    pull_s = [0.0]  # the snapshot refresh of every timed_actor call in turn, seconds: [0] = the latest

    def timed_actor(tape, iters):
        """ms per actor iteration + share of agent rows the encoder saw (an unchanged observation keeps its latent)."""
        enc = 0
        for _ in range(60):  # untimed: let the population of moving / standing agents settle under this policy (steady state: >= 50)
            actor.step(actions_override=tape_actions() if tape else None)
        # the snapshot refresh (one iteration in weights_period = 400: load_state_dict + re-packing the weight images) is timed
        # by itself below and charged at 1/400 per iteration, wherever the iteration counter happens to stand in this window
        # (rounds 2-5 left that to chance: the greedy window fell on iteration 400 and carried a whole refresh over 12 iterations)
        actor._since_pull = 1
        torch.cuda.synchronize()
        t_ = time.perf_counter()
        for _ in range(iters):
            actor.step(actions_override=tape_actions() if tape else None)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t_) / iters
        if actor.weights_period is not None:
            actor._since_pull = actor.weights_period
            t_ = time.perf_counter()
            actor.step(actions_override=tape_actions() if tape else None)
            torch.cuda.synchronize()
            pull_s[0] = max(time.perf_counter() - t_ - dt, 0.0)  # (with the latent cache: + every row encoded once again)
            pull_s.append(pull_s[0])
            dt += pull_s[0] / actor.weights_period
        if actor.latents is not None:  # (a second, untimed pass for the statistic: reading the device counter synchronises)
            for _ in range(4):
                actor.step(actions_override=tape_actions() if tape else None)
                enc += actor.latents.last_encoded()
            return dt, enc / (4.0 * E * N)
        return dt, 1.0

    # moving agents (the tape policy's actions are executed; the network's forward runs all the same), then the loop as
    # worker.py:376-414 runs it: the network's own greedy actions -- under random-init weights most agents stand still
    dt_act_tape, enc_tape = timed_actor(True, args.dqn_actor_iters)
    dt_act, enc_greedy = timed_actor(False, args.dqn_actor_iters)
    cache, actor.latents = actor.latents, None  # the same loop encoding every agent row every step
    dt_act_all, _ = timed_actor(False, args.dqn_actor_iters)
    actor.latents = cache
    if cache is not None:
        cache.key = None
    assert len(buf) >= 192 * 18, "the actor loop did not fill the replay"

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    def timed_updates():
        for _ in range(5):  # (the row counts differ from batch to batch: let the caching allocator see a few)
            learner.update()
        barrier()
        t_ = time.perf_counter()
        for _ in range(args.dqn_updates):
            learner.update()
        barrier()
        return (time.perf_counter() - t_) / args.dqn_updates

    fault_point("learner", rank)
    probe = buf.sample_batch(192)
    learner_path = "fused" if (learner._fused is not None and learner._fused.usable(probe)) else "autograd"
    from mapf_rl_amd import learner as learner_mod

    timing = world > 1 or learner_mod.FORCE_EXCHANGE
    if timing:
        learner.bucket.timing = []
    dt_upd = timed_updates()
    ex_stats = {}
    if timing:
        ex_stats = exchange_stats(learner.bucket, args.dqn_updates + 5)
        learner.bucket.timing = None
    # the same update with EVERY observation of the window through the encoder, as the reference does: no pruning of the
    # entries that cannot reach agent 0's Q-value, no reuse of repeated observations
    Network.PRUNE_UNREACHABLE, FusedUpdate.DEDUP = False, False
    learner._drop_prefetch()
    try:
        dt_upd_all = timed_updates()
    finally:
        Network.PRUNE_UNREACHABLE, FusedUpdate.DEDUP = True, True
    learner._drop_prefetch()
    reach = float(relevance(probe[7][:, :-2], probe[5]).float().mean())
    reach_min = reach_max = reach
    distinct = 1.0
    rows_enc = rows_enc_min = rows_enc_max = None
    if learner._fused is not None:
        pl = learner._fused._finish_plan(learner._fused.plan(probe))
        distinct = pl["online"].urows / max(1, pl["online"].rows)
        rows_enc = rows_enc_min = rows_enc_max = int(pl["online"].urows)  # the online encoder's batch of this rank's probe window
    # interleaved: the loop train.py runs (one update per actor iteration; the actor iteration on its own stream beside the
    # update, the replay ordered by the learner's two events -- train.py --overlap-actors, its default)
    from mapf_rl_amd.streams import role_stream

    astream = role_stream(dev, "actors")  # (one stream per role in the process: mapf_rl_amd/streams.py)

    def train_iteration(tape):
        if learner.replay_released is not None:
            astream.wait_event(learner.replay_released)
        with torch.cuda.stream(astream):
            actor.step(actions_override=tape_actions() if tape else None)
            ev = torch.cuda.Event()
            ev.record(astream)
        learner.replay_gate = ev
        learner.update()

    def timed_train(tape):
        """RAW ms per (actor iteration + update) pair.  No weights refresh falls into the window: the counter is pinned in front of it,
        as in the actor loops (round 5 corrected the figure afterwards instead -- advisor: subtracting a refresh's stand-alone cost
        understates a loop in which part of it hides beside the update)."""
        fault_point("train_loop", rank)
        astream.wait_stream(torch.cuda.current_stream(dev))
        for _ in range(60 if tape else 3):  # (tape: the population of moving agents settles again behind the greedy legs)
            with torch.cuda.stream(astream):
                actor.step(actions_override=tape_actions() if tape else None)
        torch.cuda.current_stream(dev).wait_stream(astream)
        train_iteration(tape)
        assert args.train_iters + 2 < actor.weights_period
        actor._since_pull = 1
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.train_iters):
            train_iteration(tape)
        torch.cuda.synchronize()
        learner.replay_gate = None
        if world > 1:
            dist.barrier()
        return (time.perf_counter() - t1) / args.train_iters

    dt_train = timed_train(False)
    dt_train_tape = timed_train(True)
    pull_greedy = pull_s[2] if len(pull_s) > 2 else pull_s[0]  # (the standing-policy train loop's actor iteration is the greedy one)
    pull_tape = pull_s[1] if len(pull_s) > 1 else pull_s[0]
    env.check_status()
    if world > 1:
        tt = torch.tensor([dt_upd, dt_act, dt_train, dt_upd_all, dt_act_tape, dt_act_all, dt_train_tape], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt_upd, dt_act, dt_train, dt_upd_all, dt_act_tape, dt_act_all, dt_train_tape = [float(v) for v in tt.tolist()]
        # the pruned update's encoder batch is data-dependent, hence per rank: make the spread visible
        re = float(rows_enc or 0)
        rmm = torch.tensor([reach, -reach, re, -re], dtype=torch.float64, device=dev)
        dist.all_reduce(rmm, op=dist.ReduceOp.MAX)
        reach_max, reach_min = float(rmm[0]), -float(rmm[1])
        if rows_enc is not None:  # (the time a rank reaches the collective follows its encoder batch)
            rows_enc_max, rows_enc_min = int(rmm[2]), int(-rmm[3])
        if ex_stats:  # the slowest rank's exchange figures
            keys = sorted(ex_stats)
            et = torch.tensor([ex_stats[k] for k in keys], dtype=torch.float64, device=dev)
            dist.all_reduce(et, op=dist.ReduceOp.MAX)
            ex_stats = {k: float(v) for k, v in zip(keys, et.tolist())}
    # the dominant kernel of the actor loop: the fused inference encoder (MFMA-bound), timed alone on the
    # actor's batch with HIP events on the launch stream
    obs_flat = actor.obs.reshape(E * N, 6, 9, 9)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        learner.model.encode(obs_flat)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            learner.model.encode(obs_flat)
        e1.record()
    torch.cuda.synchronize()
    enc_s = e0.elapsed_time(e1) * 1e-3 / 5
    enc_flop = 2.0 * (49 * 128 * 54 + 6 * 49 * 128 * 1152 + 49 * 16 * 128) * E * N  # 87.6 MFLOP per observation
    enc_clock = encoder_clock_ghz(learner.model, obs_flat, dev)
    P = ref_config.actor_update_steps
    out = {
        "learner_updates_per_sec": 1.0 / dt_upd, "learner_ms_per_update": dt_upd * 1e3,
        "learner_path": learner_path,
        "learner_ms_per_update_all_observations": dt_upd_all * 1e3,
        "learner_reachable_fraction": reach, "learner_reachable_fraction_min": reach_min,
        "learner_reachable_fraction_max": reach_max,
        "learner_distinct_fraction": distinct,
        "learner_rows_encoded": rows_enc, "learner_rows_encoded_min": rows_enc_min, "learner_rows_encoded_max": rows_enc_max,
        "learner_note": "only agent 0's Q-value is learned from (reference model.py:248): an update encodes the observations that can "
                        "reach it through the communication masks (learner_reachable_fraction of the window; same Q-values, "
                        "tests/test_relevance_gpu.py), and of those only the DISTINCT ones (learner_distinct_fraction: an agent that "
                        "stands still in an unchanged neighbourhood repeats its observation; same forward bits, tests/test_update_gpu.py); "
                        "learner_ms_per_update_all_observations = the same update with every observation of the window through the "
                        "encoder, as the reference does; learner_path: fused = the hand-written kernels (update.FusedUpdate), autograd = "
                        "the batch left their shape limits and ran through PyTorch",
        "learner_config": "B=192 x T=18 x A=%d windows per rank from the device replay (episodes of the actor loop under the tape policy), bf16 autocast, Adam, %s" % (
            N, "flat-bucket RCCL all-reduce x%d (synchronous data parallel: this is the job's update rate, global batch %d)" % (
                world, 192 * world) if world > 1 else "1 GPU"),
        "actor_loop_env_steps_per_sec": world * E / dt_act, "actor_loop_ms_per_iter": dt_act * 1e3,
        "actor_weights_refresh_ms": pull_greedy * 1e3 if len(pull_s) > 2 else None,
        "actor_loop_rows_encoded_fraction": enc_greedy,
        "actor_loop_tape_policy_env_steps_per_sec": world * E / dt_act_tape, "actor_loop_tape_policy_ms_per_iter": dt_act_tape * 1e3,
        "actor_loop_tape_policy_rows_encoded_fraction": enc_tape,
        "actor_loop_every_row_ms_per_iter": dt_act_all * 1e3, "actor_loop_every_row_env_steps_per_sec": world * E / dt_act_all,
        "actor_loop_note": "an agent whose 6x9x9 observation did not change since the previous step keeps its latent (the encoder is "
                           "a deterministic per-observation function: same bits, tests/test_actor_gpu.py), so the rate depends on how many "
                           "agents move: actor_loop_* = the network's own greedy actions (random-init weights: most agents stand), "
                           "actor_loop_tape_policy_* = the bench tape's 80 %% heuristic-following actions executed instead, "
                           "actor_loop_every_row_* = every agent row through the encoder every step (round 2's loop)",
        "actor_loop_config": "Network.step_batch (bf16) + mapf_step + local-buffer recording + episode flush into the device replay, %d envs x %d agents per GPU; weights snapshot pulled every %d iterations (config.actor_update_steps; timed by itself -- actor_weights_refresh_ms -- and charged at 1/%d per iteration)" % (E, N, P, P),
        # THE train-loop figure: the tape policy's actions executed by the actor (agents move, ~60 % of the rows re-encoded) beside the update
        "train_loop_tape_policy_ms_per_iter": dt_train_tape * 1e3, "train_loop_tape_policy_updates_per_sec": 1.0 / dt_train_tape,
        "train_loop_tape_policy_env_steps_per_sec": world * E / dt_train_tape,
        "train_loop_tape_policy_ms_per_iter_refresh_amortised": (dt_train_tape + pull_tape / P) * 1e3,
        # the same loop with the random-init network's own greedy actions: most agents STAND (actor_loop_rows_encoded_fraction), so the
        # actor iteration beside the update is ~1 ms -- a lower bound of the pair's cost, not what a policy that acts costs
        "train_loop_updates_per_sec": 1.0 / dt_train, "train_loop_env_steps_per_sec": world * E / dt_train,
        "train_loop_ms_per_iter": dt_train * 1e3,
        "train_loop_ms_per_iter_refresh_amortised": (dt_train + pull_greedy / P) * 1e3,
        "train_loop_config": "one actor iteration (%d envs/GPU) + one learner update per iteration, the actor iteration on its own stream beside the update "
                             "(train.py --overlap-actors); RAW wall time per pair, no weights refresh inside the window (*_refresh_amortised adds the "
                             "separately timed refresh at 1/%d).  train_loop_tape_policy_* = agents moving under the tape policy (what a policy that "
                             "acts costs); train_loop_* = standing policy (random-init greedy), the pair's lower bound" % (E, P),
        # the env-steps/s of the whole pipeline with a policy in the loop and agents moving (BASELINE metric "env steps/sec" as a
        # system rate; `value` above is the env kernel alone): the tape-policy actor loop
        "pipeline_env_steps_per_sec": world * E / dt_act_tape,
        "encoder_roofline": dict({"bound": "mfma", "kernel": "encoder_fwd_kernel", "achieved": enc_flop / enc_s / 1e12,
                                  "peak": 2500.0, "unit": "TFLOP/s", "frac": enc_flop / enc_s / 1e12 / 2500.0,
                                  "flop_per_launch": enc_flop, "kernel_avg_ms": enc_s * 1e3, "observations": E * N}, **enc_clock),
    }
    out.update({"learner_" + k: v for k, v in ex_stats.items()})
    # (leave nothing of this leg on the GPU for the next one: the replay ring alone is tens of GB at 40 agents)
    del actor, learner, buf
    import gc

    gc.collect()
    torch.cuda.empty_cache()
    return out


REF_LEVELS = [(4, 15), (3, 20), (2, 25), (6, 15), (5, 20), (1, 30), (4, 25)]  # an active-level set of a curriculum run (profiles/r02_train_curriculum_5min.log, last interval)


def ref_shape_legs(M, args, env, dev, rank, world, dist, gen):
    """The reference's OWN training configuration (config.py:25,30,50: batch 192, windows of 18 = bt_steps 16 + forward_steps 2, at most
    6 agents; worker.py:282-340) as train.py runs it: seven active curriculum levels x `--curriculum-envs` environments stepped by ONE
    captured iteration (curriculum.CurriculumActors), their episodes in one device replay laid out for 6 agents, and the learner's update
    on windows sampled from it, replayed from HIP graphs (update.FusedUpdate.GRAPH).  As in the config-2 legs the agents MOVE: the
    executed actions are the tape policy's (80 % heuristic-following, inside the captured iteration: set_policy_override) -- under the
    random-init network's own greedy actions most agents stand, few rows are re-encoded, the replay's windows are full of repeated
    observations and every figure here reads ~1.5x better than what a policy that acts costs.  Keys:
      curriculum_actor_iter_ms / _env_steps_per_sec   one actor iteration over all levels, graph-replayed, agents moving
      curriculum_actor_iter_standing_policy_ms        the same with the network's own greedy actions (random init: agents stand)
      learner_ref_shape_ms_per_update / _updates_per_sec, _graph_captures (captures inside the timed stretch: 0 = steady state)
      train_loop_ref_shape_ms_per_iter / _updates_per_sec   one such actor iteration on its own stream beside every update"""
    import torch

    import config as ref_config
    from mapf_rl_amd.curriculum import CurriculumActors
    from mapf_rl_amd.learner import Learner
    from mapf_rl_amd.model import relevance
    from mapf_rl_amd.replay import GlobalBuffer

    fault_point("ref_shape", rank)
    El, B = args.curriculum_envs, ref_config.batch_size
    torch.manual_seed(4321 + rank)
    cap = 1 << (max(2048, 2 * El * len(REF_LEVELS)) - 1).bit_length()  # train.py's default capacity rule
    buf = GlobalBuffer(cap, max_agents=ref_config.max_num_agetns, device=dev, init_set=ref_config.init_set,
                       max_map_length=ref_config.max_map_lenght, pass_rate=ref_config.pass_rate)
    buf.stat_dict = {k: [] for k in REF_LEVELS}
    torch.manual_seed(1234)  # identical initial weights on every rank
    learner = Learner(buf, device=dev, batch_size=B)
    torch.manual_seed(4321 + rank)
    P = ref_config.actor_update_steps
    cur = CurriculumActors(learner.model, buf, envs_per_level=El, device=dev, seed=3 + rank, reward_fn=ref_config.reward_fn, weights_period=P)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    iters = args.curriculum_iters

    def timed_iterations():
        for _ in range(40):  # (the first iterations are issued directly, then captured; the population of moving agents settles)
            cur.step()
        cur._since_pull = 1  # (no weights refresh inside a timed stretch)
        assert iters + 2 < P
        r0 = cur.graph_replays
        barrier()
        t_ = time.perf_counter()
        for _ in range(iters):
            cur.step()
        t_host = time.perf_counter() - t_
        barrier()
        rows = cur.latents.last_encoded() / max(1, cur.obs_all.shape[0]) if cur.latents is not None else 1.0
        return (time.perf_counter() - t_) / iters, t_host / iters, cur.graph_replays - r0, rows

    dt_cur_stand, _, _, rows_stand = timed_iterations()
    cur.set_policy_override(heuristic_actions_rows)
    # episodes enter the replay when they end: under the tape policy most finish (all agents on their goals) within ~100 steps
    for _ in range(300):
        cur.step()
    dt_cur, host_cur, replayed, rows_tape = timed_iterations()
    torch.cuda.synchronize()
    assert len(buf) >= B * 18, "the curriculum actors did not fill the replay"
    # ---- the learner's update at this shape ----
    fu = learner._fused
    cur._since_pull = 1
    for _ in range(args.ref_shape_warmup):  # (graph mode: the (rows, distinct rows) buckets this replay produces get captured)
        learner.update()
    barrier()
    c0, g0 = (fu.graph_captures, fu.graph_replays) if fu is not None else (0, 0)
    U = args.ref_shape_updates
    from mapf_rl_amd import learner as learner_mod

    timing = world > 1 or learner_mod.FORCE_EXCHANGE
    if timing:
        learner.bucket.timing = []
    t_ = time.perf_counter()
    for _ in range(U):
        learner.update()
    t_host_u = time.perf_counter() - t_
    barrier()
    dt_upd = (time.perf_counter() - t_) / U
    ex_stats = {}
    if timing:  # (graph mode: the whole exchange follows the captured backward stage -- nothing of it is hidden behind the update here)
        ex_stats = exchange_stats(learner.bucket, U)
        learner.bucket.timing = None
    caps = (fu.graph_captures - c0) if fu is not None else None
    reps = (fu.graph_replays - g0) if fu is not None else None
    learner._drop_prefetch()
    probe = buf.sample_batch(B)
    path = "fused" if (fu is not None and fu.usable(probe)) else "autograd"
    reach = float(relevance(probe[7][:, :-2], probe[5]).float().mean())
    rows_enc = None
    if fu is not None:
        pl = fu._finish_plan(fu.plan(probe))
        rows_enc = (int(pl["online"].urows), int(pl["online"].rows))
    # ---- the pair train.py runs: one curriculum actor iteration on its own stream beside every update ----
    from mapf_rl_amd.streams import role_stream

    astream = role_stream(dev, "actors")  # (one stream per role in the process: mapf_rl_amd/streams.py)

    def train_iteration():
        if learner.replay_released is not None:
            astream.wait_event(learner.replay_released)
        with torch.cuda.stream(astream):
            cur.step()
            ev = torch.cuda.Event()
            ev.record(astream)
        learner.replay_gate = ev
        learner.update()

    astream.wait_stream(torch.cuda.current_stream(dev))
    for _ in range(10):
        train_iteration()
    assert U + 12 < P
    cur._since_pull = 1
    barrier()
    c1 = fu.graph_captures if fu is not None else 0
    t_ = time.perf_counter()
    for _ in range(U):
        train_iteration()
    torch.cuda.synchronize()
    learner.replay_gate = None
    if world > 1:
        dist.barrier()
    dt_train = (time.perf_counter() - t_) / U
    caps_train = (fu.graph_captures - c1) if fu is not None else None
    if world > 1:
        tt = torch.tensor([dt_cur, dt_upd, dt_train, dt_cur_stand], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt_cur, dt_upd, dt_train, dt_cur_stand = [float(v) for v in tt.tolist()]
        if ex_stats:  # the slowest rank's exchange figures
            keys = sorted(ex_stats)
            et = torch.tensor([ex_stats[k] for k in keys], dtype=torch.float64, device=dev)
            dist.all_reduce(et, op=dist.ReduceOp.MAX)
            ex_stats = {k: float(v) for k, v in zip(keys, et.tolist())}
    steps_per_iter = world * El * len(REF_LEVELS)
    out = {
        "curriculum_actor_iter_ms": dt_cur * 1e3, "curriculum_actor_iter_host_ms": host_cur * 1e3,
        "curriculum_actor_env_steps_per_sec": steps_per_iter / dt_cur, "curriculum_actor_graph_replays": replayed,
        "curriculum_actor_rows_encoded_fraction": rows_tape,
        "curriculum_actor_iter_standing_policy_ms": dt_cur_stand * 1e3, "curriculum_actor_standing_policy_rows_encoded_fraction": rows_stand,
        "curriculum_actor_config": "%d levels %s x %d envs per level per GPU, all levels stepped by one captured iteration, %d timed iterations; agents move "
                                   "under the tape policy (80 %% heuristic-following, drawn inside the captured iteration); *_standing_policy_* = the random-init "
                                   "network's own greedy actions" % (len(REF_LEVELS), REF_LEVELS, El, iters),
        "learner_ref_shape_ms_per_update": dt_upd * 1e3, "learner_ref_shape_updates_per_sec": 1.0 / dt_upd,
        "learner_ref_shape_host_ms_per_update": t_host_u / U * 1e3,
        "learner_ref_shape_graph_captures": caps, "learner_ref_shape_graph_replays": reps, "learner_ref_shape_path": path,
        "learner_ref_shape_reachable_fraction": reach, "learner_ref_shape_rows_encoded": rows_enc[0] if rows_enc else None,
        "learner_ref_shape_rows": rows_enc[1] if rows_enc else None,
        "learner_ref_shape_config": "B=%d x T=18 x A<=%d windows per rank from the curriculum actors' replay (episodes under the tape policy), bf16, graph replay %s, "
                                    "%d warm-up + %d timed updates%s" % (B, ref_config.max_num_agetns, "on" if (fu is not None and fu.graph_mode()) else "off",
                                                                        args.ref_shape_warmup, U, ", gradient exchange x%d" % world if world > 1 else ""),
        "train_loop_ref_shape_ms_per_iter": dt_train * 1e3, "train_loop_ref_shape_updates_per_sec": 1.0 / dt_train,
        "train_loop_ref_shape_env_steps_per_sec": steps_per_iter / dt_train, "train_loop_ref_shape_graph_captures": caps_train,
        "train_loop_ref_shape_config": "one curriculum actor iteration (its own stream, agents moving under the tape policy) + one update per iteration: the loop of "
                                       "`python train.py` once training started",
    }
    out.update({"learner_ref_shape_" + k: v for k, v in ex_stats.items()})
    del cur, learner, buf
    return out


if __name__ == "__main__":
    main()
