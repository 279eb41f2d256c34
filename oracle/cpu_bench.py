"""TEST INFRASTRUCTURE ONLY -- the `cpu_baseline` leg of bench.py, run as child processes that never import
torch and never touch the GPU.

    python -m oracle.cpu_bench <inputs.npz> <seconds>  ->  one JSON line on stdout

inputs.npz: maps int8 [S,L,L], agents int16 [S,N,2], goals int16 [S,N,2], tape int8 [T,S,N].

SURVEY.md 8(d): the reference runs one single-threaded actor per CPU core (`OMP_NUM_THREADS=1`, reference
train.py:2,23; worker.py:355).  Mirrored here: one single-threaded oracle process per usable CPU, each pinned to
its own CPU with sched_setaffinity, each stepping its own share of the sample environments (sequential reference
semantics, step + observe every step) for the same wall-clock window; the rate is the sum over workers.  "Usable"
= the CPUs in this process's affinity mask, capped by the cgroup CPU quota when the container has one (a GPU box
hands each GPU a share of the host, e.g. 16 of 256 logical CPUs: more runnable processes than that only
time-slice).  A short scan over worker counts (quota, 2x quota, all CPUs) keeps the fastest and reports all.
Also reports the CPU model, logical / physical core counts and the final positions of the first pass so that
the caller can assert GPU/CPU trajectory identity."""
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle  # noqa: E402


def cpu_info():
    model, phys = "unknown", set()
    try:
        pid = cid = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name") and model == "unknown":
                model = line.split(":", 1)[1].strip()
            elif line.startswith("physical id"):
                pid = line.split(":", 1)[1].strip()
            elif line.startswith("core id"):
                cid = line.split(":", 1)[1].strip()
            elif not line.strip():
                if pid is not None and cid is not None:
                    phys.add((pid, cid))
                pid = cid = None
    except OSError:
        pass
    return model, len(phys)


def cpu_quota():
    """cgroup CPU quota in cores (None = unlimited / unknown)."""
    try:  # cgroup v2
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            return float(q) / float(p)
    except (OSError, ValueError):
        pass
    try:  # cgroup v1
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0:
            return q / p
    except (OSError, ValueError):
        pass
    return None


def _worker(cpu, path, lo, hi, t_begin, seconds, out_q):
    try:
        os.sched_setaffinity(0, {cpu})
    except OSError:
        pass
    z = np.load(path)
    maps, agents, goals, tape = z["maps"][lo:hi], z["agents"][lo:hi], z["goals"][lo:hi], np.ascontiguousarray(z["tape"][:, lo:hi])
    nv = oracle.navi_batch(maps, goals, 1)
    kw = dict(want_pos=False, want_rclass=False, want_done=False, want_hash=True, nthreads=1)
    first = oracle.rollout(maps, agents, goals, nv, tape, **kw)  # warm-up + the trajectory check
    while time.time() < t_begin:  # all workers start their timed window together
        time.sleep(0.001)
    t0 = time.perf_counter()
    reps = 0
    while time.perf_counter() - t0 < seconds:
        oracle.rollout(maps, agents, goals, nv, tape, **kw)
        reps += 1
    dt = time.perf_counter() - t0
    out_q.put((lo, hi, reps, dt, first["status"], first["final_agents"]))


def run(path, S, T, cpus, seconds):
    """`len(cpus)` pinned single-threaded workers over S environments; returns (env-steps/s, final positions [S,N,2])."""
    n = min(len(cpus), S)
    bounds = [S * k // n for k in range(n + 1)]
    q = mp.Queue()
    t_begin = time.time() + 1.0 + 0.002 * n
    procs = [mp.Process(target=_worker, args=(cpus[k], path, bounds[k], bounds[k + 1], t_begin, seconds, q)) for k in range(n)]
    for p in procs:
        p.start()
    res = [q.get() for _ in procs]
    for p in procs:
        p.join()
    assert all(r[4] == 0 for r in res), "oracle status"
    rate = sum((hi - lo) * T * reps / dt for lo, hi, reps, dt, _, _ in res)
    res.sort(key=lambda r: r[0])
    return rate, np.concatenate([r[5] for r in res], axis=0)


def main():
    path = sys.argv[1]
    seconds = float(sys.argv[2])
    z = np.load(path)
    S, T = z["maps"].shape[0], z["tape"].shape[0]
    cpus = sorted(os.sched_getaffinity(0))
    quota = cpu_quota()
    model, phys = cpu_info()
    cands = {len(cpus)}
    if quota is not None:
        cands |= {max(1, int(quota + 0.5)), min(len(cpus), max(1, int(2 * quota + 0.5)))}
    else:
        cands |= {max(1, len(cpus) // 2)}
    scan, final = {}, None
    for c in sorted(cands):
        scan[c], final = run(path, S, T, cpus[:c], 1.5)
    best = max(scan, key=scan.get)
    rate, final = run(path, S, T, cpus[:best], seconds)
    how = "affinity mask %d CPUs, cgroup quota %s" % (len(cpus), "none" if quota is None else "%.1f cores" % quota)
    print(json.dumps({"env_steps_per_sec": rate, "seconds": seconds, "workers": best, "how": how, "cpu_model": model,
                      "logical_cpus": os.cpu_count(), "physical_cores": phys, "cpu_quota_cores": quota, "envs": int(S),
                      "tape_steps": int(T), "scan": {str(k): v for k, v in scan.items()}, "final_agents": final.tolist()}))


if __name__ == "__main__":
    main()
