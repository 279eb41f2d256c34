"""TEST INFRASTRUCTURE ONLY -- ctypes binding of oracle/libmapf_oracle.so (mapf_oracle.c).

The oracle restates, with the reference's sequential semantics, Environment.get_navi_map /
step / observe (reference environment.py:217-467).  It is the parity checker and the `cpu_baseline`
of bench.py; it is never the thing shipped or measured as the product.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libmapf_oracle.so")

REWARD_TABLE = np.array([-0.075, 0.0, -0.075, -0.5, 3.0], dtype=np.float64)  # reference config.py:8-12
RC_MOVE, RC_STAY_ON, RC_STAY_OFF, RC_COLLISION, RC_FINISH = range(5)
ERR_ACTION, ERR_UNIQUE, ERR_IDCHECK = -1, -2, -3


def build(force=False):
    src = os.path.join(_HERE, "mapf_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "libmapf_oracle.so"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        c_int, vp = ctypes.c_int, ctypes.c_void_p
        L.oracle_navi.argtypes = [c_int, c_int, vp, vp, vp]
        L.oracle_navi.restype = None
        L.oracle_dist.argtypes = [c_int, vp, c_int, c_int, vp]
        L.oracle_dist.restype = None
        L.oracle_step.argtypes = [c_int, c_int, vp, vp, vp, vp, vp, vp]
        L.oracle_step.restype = c_int
        L.oracle_observe.argtypes = [c_int, c_int, c_int, vp, vp, vp, vp]
        L.oracle_observe.restype = None
        L.oracle_rollout.argtypes = [c_int] * 5 + [vp] * 10 + [c_int]
        L.oracle_rollout.restype = c_int
        L.oracle_navi_batch.argtypes = [c_int, c_int, c_int, vp, vp, vp, c_int]
        L.oracle_navi_batch.restype = None
        L.oracle_max_threads.restype = c_int
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _c(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


def navi(map_, goals):
    """map [L,L], goals [N,2] -> uint8 [N,4,L,L] (unpadded navi flags)."""
    map_ = _c(map_, np.int8)
    goals = _c(goals, np.int16)
    L, N = map_.shape[0], goals.shape[0]
    out = np.zeros((N, 4, L, L), dtype=np.uint8)
    lib().oracle_navi(L, N, _p(map_), _p(goals), _p(out))
    return out


def dist(map_, goal):
    map_ = _c(map_, np.int8)
    L = map_.shape[0]
    out = np.zeros((L, L), dtype=np.int32)
    lib().oracle_dist(L, _p(map_), int(goal[0]), int(goal[1]), _p(out))
    return out


def step(map_, agents, goals, actions):
    """Returns (status, new_agents int16[N,2], rclass int8[N], done bool)."""
    map_ = _c(map_, np.int8)
    agents = _c(agents, np.int16).copy()
    goals = _c(goals, np.int16)
    actions = _c(actions, np.int8)
    L, N = map_.shape[0], agents.shape[0]
    assert actions.shape == (N,)
    rclass = np.zeros(N, dtype=np.int8)
    done = np.zeros(1, dtype=np.uint8)
    st = lib().oracle_step(L, N, _p(map_), _p(agents), _p(goals), _p(actions), _p(rclass), _p(done))
    return st, agents, rclass, bool(done[0])


def observe(map_, agents, navi_, r=4):
    map_ = _c(map_, np.int8)
    agents = _c(agents, np.int16)
    navi_ = _c(navi_, np.uint8)
    L, N = map_.shape[0], agents.shape[0]
    W = 2 * r + 1
    obs = np.zeros((N, 6, W, W), dtype=np.uint8)
    lib().oracle_observe(L, N, r, _p(map_), _p(agents), _p(navi_), _p(obs))
    return obs


def navi_batch(maps, goals, nthreads=0):
    maps = _c(maps, np.int8)
    goals = _c(goals, np.int16)
    E, L = maps.shape[0], maps.shape[1]
    N = goals.shape[1]
    out = np.zeros((E, N, 4, L, L), dtype=np.uint8)
    lib().oracle_navi_batch(E, L, N, _p(maps), _p(goals), _p(out), nthreads)
    return out


def rollout(maps, agents, goals, navi_, tape, r=4, want_pos=True, want_rclass=True, want_done=True,
            want_obs_last=False, want_hash=False, nthreads=0):
    """E envs x T steps from an action tape int8[T,E,N].  Returns dict of outputs; `agents` is not modified."""
    maps = _c(maps, np.int8)
    agents = _c(agents, np.int16).copy()
    goals = _c(goals, np.int16)
    navi_ = _c(navi_, np.uint8)
    tape = _c(tape, np.int8)
    E, L = maps.shape[0], maps.shape[1]
    N = goals.shape[1]
    T = tape.shape[0]
    assert tape.shape == (T, E, N)
    W = 2 * r + 1
    pos = np.zeros((T, E, N, 2), dtype=np.int16) if want_pos else None
    rc = np.zeros((T, E, N), dtype=np.int8) if want_rclass else None
    dn = np.zeros((T, E), dtype=np.uint8) if want_done else None
    ol = np.zeros((E, N, 6, W, W), dtype=np.uint8) if want_obs_last else None
    hs = np.zeros((T, E), dtype=np.uint64) if want_hash else None
    st = lib().oracle_rollout(E, L, N, r, T, _p(maps), _p(agents), _p(goals), _p(navi_), _p(tape),
                              _p(pos), _p(rc), _p(dn), _p(ol), _p(hs), nthreads)
    return dict(status=st, pos=pos, rclass=rc, done=dn, obs_last=ol, obs_hash=hs, final_agents=agents)


def max_threads():
    return lib().oracle_max_threads()
