#!/usr/bin/env python3
"""TEST / MEASUREMENT INFRASTRUCTURE ONLY (build container; /root/reference does not exist on the GPU box).

Times the UNMODIFIED reference's `Environment.step` (reference environment.py:278-430, which ends in `observe()`, :433-467) at
BASELINE configs[1]'s shape -- 32x32 grid, 40 agents, obstacle density 0.3 -- on one host core, with warm-up and repeats, as
BASELINE.md section 3.1 planned ("to be re-measured with warm-up and >= 5 repeats when the oracle harness exists").

    python -m oracle.time_reference [--repeats 7] [--steps 1000] > profiles/r05_reference_env_step_cpu.txt
    python -m oracle.time_reference --fixture 16          # BASELINE configs[0], reading C1a: the 200 cases of test16_40_0.3.pkl (40x40, 16 agents)
    python -m oracle.time_reference --map 16 --agents 40  # ... reading C1b: BASELINE's literal "16x16 grid, 40 agents"

With --oracle (default on) the C restatement (oracle/mapf_oracle.c: step + observe, one thread) is timed on the same scenarios under the
same policy beside the reference, so that a configs[0] table has reference-CPU and oracle-CPU figures from one box.

Scenarios: Bernoulli(0.3) maps with the reference's placement rule (this repository's generator at fixed density; loaded through the
reference's own `Environment.load`).  Two action policies: the bench's tape policy (80 % follow a navigation flag of the own cell,
20 % uniform) and uniform random (what the survey's single-run figure used).  Only `env.step(actions)` is inside the
timed region; choosing the actions and reloading a scenario when an episode ends (done, or 256 steps: worker.py:390) are outside."""
import argparse
import os
import platform
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--map", type=int, default=32)
    ap.add_argument("--agents", type=int, default=40)
    ap.add_argument("--density", type=float, default=0.3)
    ap.add_argument("--warmup", type=int, default=300)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--repeats", type=int, default=7)
    ap.add_argument("--fixture", type=int, default=0, help="16 / 32 / 64: the scenarios of the reference's test{N}_40_0.3.pkl (committed copy: "
                    "tests/golden/fixture_scenarios.npz) instead of generated ones; --map / --agents follow the fixture")
    ap.add_argument("--no-oracle", action="store_true", help="skip the C oracle's leg")
    a = ap.parse_args()
    os.environ.setdefault("OMP_NUM_THREADS", "1")  # reference train.py:2
    import numpy as np

    from oracle import ref_harness

    ref = ref_harness.load_reference()
    import mapf_rl_amd as M

    L, N = a.map, a.agents
    if a.fixture:
        from mapf_rl_amd.evaluate import load_fixture_npz

        t = load_fixture_npz(os.path.join(ROOT, "tests", "golden", "fixture_scenarios.npz"), a.fixture)
        maps, agents, goals = np.stack(t["maps"]).astype(np.int8), np.stack(t["agents"]).astype(np.int16), np.stack(t["goals"]).astype(np.int16)
        L, N = maps.shape[1], agents.shape[1]
        what = "the %d cases of test%d_40_0.3.pkl" % (maps.shape[0], a.fixture)
    else:
        maps, agents, goals, redraws = M.generate_scenarios(64, L, N, a.density, seed=2024)
        what = "64 generated scenarios (Bernoulli(%.2f) maps, the reference's placement rule; %d infeasible draws skipped)" % (a.density, redraws)
    rng = np.random.RandomState(0)

    def run(policy):
        env = ref.environment.Environment(num_agents=N, map_length=L)
        k = [0]

        def reload():
            e = k[0] % maps.shape[0]
            k[0] += 1
            env.load(maps[e].astype(np.float32), agents[e].astype(int), goals[e].astype(int))
            return env.observe()

        def choose(obs):
            uni = rng.randint(0, 5, N)
            if policy == "uniform":
                return uni.tolist()
            flags = obs[0][:, 2:6, 4, 4] != 0            # navigation flags of the own cell (environment.py:455-465)
            score = rng.random_sample((N, 4)) * flags
            follow = np.where(flags.any(1), 1 + score.argmax(1), 0)
            return np.where(rng.random_sample(N) < 0.8, follow, uni).tolist()

        obs = reload()
        rates, t_step = [], []
        for rep in range(a.repeats + 1):  # repeat 0 = warm-up
            n = a.warmup if rep == 0 else a.steps
            spent = 0.0
            for _ in range(n):
                act = choose(obs)
                t0 = time.perf_counter()
                obs, _, done, _ = env.step(act)
                spent += time.perf_counter() - t0
                if done or env.steps >= 256:
                    obs = reload()
            if rep:
                rates.append(n / spent)
                t_step.append(spent / n * 1e3)
        return rates, t_step

    cpu = platform.processor() or "?"
    try:
        cpu = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        pass
    print("reference Environment.step (incl. observe), %dx%d grid, %d agents, %s; 1 core of %s; python %s, numpy %s" % (
        L, L, N, what, cpu, platform.python_version(), np.__version__))
    print("warm-up %d steps, then %d repeats of %d steps; only env.step() is timed" % (a.warmup, a.repeats, a.steps))
    for policy in ("tape (80 % heuristic / 20 % uniform)", "uniform"):
        rates, ms = run(policy.split(" ")[0])
        print("policy %-38s env-steps/s median %.0f  min %.0f  max %.0f   (ms per step: median %.3f)   repeats: %s" % (
            policy, statistics.median(rates), min(rates), max(rates), statistics.median(ms), " ".join("%.0f" % r for r in rates)))
    if not a.no_oracle:
        oracle_leg(a, np, maps, agents, goals, L, N)


def oracle_leg(a, np, maps, agents, goals, L, N):
    """The C oracle (sequential reference semantics, step + observe every step) on ONE environment at a time, one thread: an action tape
    per scenario drawn by the tape policy from the oracle's own observations, then the tape replayed `repeats` times under the clock."""
    from oracle import oracle

    rng = np.random.RandomState(1)
    T = 64
    S = min(16, maps.shape[0])
    tapes, navis = [], []
    for e in range(S):
        nv = oracle.navi(maps[e], goals[e])
        pos = agents[e].copy()
        tape = np.zeros((T, 1, N), dtype=np.int8)
        for t in range(T):
            obs = oracle.observe(maps[e], pos, nv)
            flags = obs[:, 2:6, 4, 4] != 0
            score = rng.random_sample((N, 4)) * flags
            follow = np.where(flags.any(1), 1 + score.argmax(1), 0)
            tape[t, 0] = np.where(rng.random_sample(N) < 0.8, follow, rng.randint(0, 5, N))
            st, pos, _, _ = oracle.step(maps[e], pos, goals[e], tape[t, 0])
            assert st == 0
        tapes.append(tape)
        navis.append(nv[None])
    kw = dict(want_pos=False, want_rclass=False, want_done=False, want_hash=True, nthreads=1)
    rates = []
    for rep in range(a.repeats + 1):
        t0 = time.perf_counter()
        n = 0
        while n < a.steps * 4:
            for e in range(S):
                oracle.rollout(maps[e:e + 1], agents[e:e + 1], goals[e:e + 1], navis[e], tapes[e], **kw)
                n += T
        if rep:
            rates.append(n / (time.perf_counter() - t0))
    print("C oracle (oracle/mapf_oracle.c, step + observe, ONE environment at a time, 1 thread), same shape, tape policy: env-steps/s median %.0f  min %.0f  max %.0f "
          "(us per step: median %.1f)" % (statistics.median(rates), min(rates), max(rates), 1e6 / statistics.median(rates)))


if __name__ == "__main__":
    main()
