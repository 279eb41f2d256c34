#!/usr/bin/env python3
"""The pair train.py runs at the reference's training shape -- one curriculum actor iteration (agents moving under the tape policy) on its
own stream beside every graph-replayed update -- under different stream arrangements:
  MODE=base      actors on a normal-priority stream, learner on the default stream + a normal second stream (train.py today)
  MODE=lprio     learner on a HIGH-priority stream (+ high-priority second stream), actors normal
  MODE=alow      learner normal, actors on a LOW-priority HIP stream (hipStreamCreateWithPriority, through torch.cuda.ExternalStream)
  MODE=mask:<n>  actors on a stream restricted to the first <n> CUs of every XCD pair ... (hipExtStreamCreateWithCUMask)
  MODE=serial    no second stream for the actors: strictly alternating
  MODE=free      the actors do not flush into the replay and are not ordered against the update at all (the potential of splitting
                 the captured iteration in front of its episode flush)
Usage: MODE=... python tools/micro/train_loop_overlap.py [envs_per_level] [pairs]"""
import ctypes
import os
import sys
import time

MODE = os.environ.get("MODE", "base")
if MODE == "lprio":
    os.environ["MAPF_LEARNER_STREAM_PRIORITY"] = "-1"
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import config as ref_config  # noqa: E402
from bench import REF_LEVELS, heuristic_actions_rows  # noqa: E402
from mapf_rl_amd.curriculum import CurriculumActors  # noqa: E402
from mapf_rl_amd.learner import Learner  # noqa: E402
from mapf_rl_amd.replay import GlobalBuffer  # noqa: E402

El = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
U = int(sys.argv[2]) if len(sys.argv) > 2 else 300
dev = torch.device("cuda")
hip = ctypes.CDLL("libamdhip64.so")


def hip_stream(priority=None, cu_mask=None):
    s = ctypes.c_void_p()
    if cu_mask is not None:
        words = (ctypes.c_uint32 * len(cu_mask))(*cu_mask)
        rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), len(cu_mask), words)
    else:
        rc = hip.hipStreamCreateWithPriority(ctypes.byref(s), 1, priority)  # 1 = hipStreamNonBlocking
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value, device=dev)


lo, hi = ctypes.c_int(), ctypes.c_int()
hip.hipDeviceGetStreamPriorityRange(ctypes.byref(lo), ctypes.byref(hi))
main = torch.cuda.Stream(device=dev, priority=-1) if MODE == "lprio" else torch.cuda.current_stream(dev)
with torch.cuda.stream(main):
    torch.manual_seed(4321)
    cap = 1 << (max(2048, 2 * El * len(REF_LEVELS)) - 1).bit_length()
    buf = GlobalBuffer(cap, max_agents=6, device=dev, init_set=ref_config.init_set, max_map_length=40, pass_rate=0.9)
    buf.stat_dict = {k: [] for k in REF_LEVELS}
    learner = Learner(buf, device=dev, batch_size=192)
    cur = CurriculumActors(learner.model, buf, envs_per_level=El, device=dev, seed=3, reward_fn=ref_config.reward_fn, weights_period=400)
    cur.set_policy_override(heuristic_actions_rows)
    for _ in range(340):
        cur.step()
    for _ in range(80):
        learner.update()
    torch.cuda.synchronize()
    if MODE == "serial":
        astream = None
    elif MODE == "alow":
        astream = hip_stream(priority=lo.value)
    elif MODE.startswith("mask:"):
        n = int(MODE.split(":")[1])  # CUs (of 256) the actors may use: the first n bits of the mask
        words = [(0xFFFFFFFF if n >= 32 * (k + 1) else ((1 << max(0, n - 32 * k)) - 1)) for k in range(8)]
        astream = hip_stream(cu_mask=words)
    else:
        from mapf_rl_amd.streams import role_stream
        astream = role_stream(dev, "actors")

    if MODE == "free":
        # the POTENTIAL of an actor iteration that is not ordered against the learner's replay operations at all: the actors stop
        # flushing episodes (the learner keeps sampling what the replay holds), so nothing needs ordering
        cur.buffer = None
        cur._graph, cur._warm = None, 0

    def pair():
        if MODE == "free":
            with torch.cuda.stream(astream):
                cur.step()
            learner.update()
            return
        if astream is None:
            cur.step()
            learner.update()
            return
        if learner.replay_released is not None:
            astream.wait_event(learner.replay_released)
        with torch.cuda.stream(astream):
            cur.step()
            ev = torch.cuda.Event()
            ev.record(astream)
        learner.replay_gate = ev
        learner.update()

    if astream is not None:
        astream.wait_stream(main)
    for _ in range(30):
        pair()
    cur._since_pull = 1
    torch.cuda.synchronize()
    c0 = learner._fused.graph_captures
    t0 = time.perf_counter()
    for _ in range(U):
        pair()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / U
    # the halves alone, for reference
    t0 = time.perf_counter()
    for _ in range(100):
        learner.update()
    torch.cuda.synchronize()
    du = (time.perf_counter() - t0) / 100
    cur._since_pull = 1
    t0 = time.perf_counter()
    for _ in range(100):
        cur.step()
    torch.cuda.synchronize()
    da = (time.perf_counter() - t0) / 100
print("MODE=%-9s priority range (least %d, greatest %d): pair %.3f ms = %.1f updates/s (captures in the stretch %d); update alone %.3f ms, actor iteration alone %.3f ms, sum %.3f" % (
    MODE, lo.value, hi.value, dt * 1e3, 1 / dt, learner._fused.graph_captures - c0, du * 1e3, da * 1e3, (du + da) * 1e3), flush=True)
