"""Host-side mirror of the reference's `Environment` (reference environment.py:74-467) on top of the
C ABI in include/mapf_env.h.

* `VecEnvironment` -- E lock-step environments resident on one MI355X; tensors in, tensors out,
  launches on torch's current HIP stream.  This is the product interface for training / benchmarking.
* `Environment`    -- source-compatible single-environment facade (same constructor, `reset`, `load`,
  `step`, `observe` signatures, return shapes/dtypes and attributes as the reference), implemented as a
  1-env VecEnvironment.  Errors map back to the reference's exceptions (AssertionError for bad actions,
  RuntimeError('unique'), ValueError when placement runs out of cells).

PyTorch is used for device memory and streams only.
"""
import ctypes
import random
from typing import List

import numpy as np
import torch

from . import _lib
from ._lib import check, lib

# reference environment.py:12
action_list = np.array([[0, 0], [-1, 0], [1, 0], [0, -1], [0, 1]], dtype=np.int64)

# reference config.py:8-12, in MAPF_RC_* order
REWARD_KEYS = ("move", "stay_on_goal", "stay_off_goal", "collision", "finish")
DEFAULT_REWARD_FN = dict(move=-0.075, stay_on_goal=0, stay_off_goal=-0.075, collision=-0.5, finish=3)
OBS_RADIUS = 4


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def generate_scenarios(num_envs, map_length, num_agents, density=-1.0, seed=0):
    """Host scenario generator (reference rule, environment.py:100-138).  Returns numpy
    (maps int8[E,L,L], agents int16[E,N,2], goals int16[E,N,2], redraws)."""
    maps = np.zeros((num_envs, map_length, map_length), np.int8)
    agents = np.zeros((num_envs, num_agents, 2), np.int16)
    goals = np.zeros((num_envs, num_agents, 2), np.int16)
    redraws = ctypes.c_int32(0)
    st = lib.mapf_generate(num_envs, map_length, num_agents, float(density), int(seed),
                           maps.ctypes.data, agents.ctypes.data, goals.ctypes.data, ctypes.byref(redraws))
    if st == _lib.ERR_NO_SPACE:
        raise ValueError("no empty position left for %d agents on a %dx%d map" % (num_agents, map_length, map_length))
    check(st, "mapf_generate")
    return maps, agents, goals, redraws.value


class VecEnvironment:
    """E independent MAPF environments of one shape, stepped in lock-step on one GPU."""

    def __init__(self, num_envs, map_length, num_agents, obs_radius=OBS_RADIUS, reward_fn=None, device=None):
        if not torch.cuda.is_available():
            raise RuntimeError("mapf_rl_amd.VecEnvironment needs a HIP device (no CPU fallback)")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("mapf_rl_amd.VecEnvironment needs a HIP device (no CPU fallback)")
        self.num_envs, self.map_length, self.num_agents, self.obs_radius = num_envs, map_length, num_agents, obs_radius
        self._h = ctypes.c_void_p()
        dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.device = torch.device("cuda", dev_index)
        check(lib.mapf_create(num_envs, map_length, num_agents, obs_radius, dev_index, ctypes.byref(self._h)),
              "mapf_create")
        self.reward_fn = dict(DEFAULT_REWARD_FN if reward_fn is None else reward_fn)
        tab = (ctypes.c_float * 5)(*[float(self.reward_fn[k]) for k in REWARD_KEYS])
        check(lib.mapf_set_reward_table(self._h, tab), "mapf_set_reward_table")
        E, N, W = num_envs, num_agents, 2 * obs_radius + 1
        d = self.device
        # persistent output buffers (re-used every step; callers that keep history must clone)
        self.obs = torch.empty((E, N, 6, W, W), dtype=torch.uint8, device=d)
        self.pos = torch.empty((E, N, 2), dtype=torch.int16, device=d)
        self.reward_class = torch.empty((E, N), dtype=torch.int8, device=d)
        self.reward = torch.empty((E, N), dtype=torch.float32, device=d)
        self.done = torch.empty((E,), dtype=torch.uint8, device=d)
        self.obs_bits_row_dwords = lib.mapf_obs_bits_row_dwords(self._h)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value and lib is not None:
            lib.mapf_destroy(h)
            self._h = ctypes.c_void_p()

    # -- scenario ingestion (reference Environment.load, environment.py:198-215) --
    def load(self, maps, agents_pos, goals_pos):
        """maps [E,L,L], agents_pos/goals_pos [E,N,2]; numpy arrays (host) or torch tensors on this device."""
        E, L, N = self.num_envs, self.map_length, self.num_agents
        if isinstance(maps, torch.Tensor):
            m = maps.to(self.device, torch.int8).contiguous()
            a = agents_pos.to(self.device, torch.int16).contiguous()
            g = goals_pos.to(self.device, torch.int16).contiguous()
            assert m.shape == (E, L, L) and a.shape == (E, N, 2) and g.shape == (E, N, 2)
            check(lib.mapf_load(self._h, _ptr(m), _ptr(a), _ptr(g), 1, _stream(self.device)), "mapf_load")
            torch.cuda.current_stream(self.device).synchronize()  # m/a/g may be temporaries
        else:
            m = np.ascontiguousarray(np.asarray(maps) != 0, dtype=np.int8)
            a = np.ascontiguousarray(agents_pos, dtype=np.int16)
            g = np.ascontiguousarray(goals_pos, dtype=np.int16)
            assert m.shape == (E, L, L) and a.shape == (E, N, 2) and g.shape == (E, N, 2)
            check(lib.mapf_load(self._h, m.ctypes.data, a.ctypes.data, g.ctypes.data, 0, _stream(self.device)),
                  "mapf_load")
        check(lib.mapf_build_navi(self._h, _stream(self.device)), "mapf_build_navi")

    def reset(self, density=-1.0, seed=0):
        """Fresh random scenarios by the reference's rule (environment.py:146-196); returns (obs, pos)."""
        maps, agents, goals, _ = generate_scenarios(self.num_envs, self.map_length, self.num_agents, density, seed)
        self.load(maps, agents, goals)
        return self.observe()

    def reset_envs(self, mask=None, density=-1.0, seed=0):
        """On-device reset of the environments flagged in `mask` (uint8/bool tensor [E]; None = all): new map,
        placement, navi fields, steps = 0.  Asynchronous, no host round trip."""
        m = None
        if mask is not None:
            m = mask.to(self.device, torch.uint8).contiguous()
            assert m.shape == (self.num_envs,)
        check(lib.mapf_reset_envs(self._h, _ptr(m), float(density), int(seed) & 0xFFFFFFFFFFFFFFFF, _stream(self.device)),
              "mapf_reset_envs")
        self._keep_mask = m

    def stage_next(self, density=-1.0, seed=0):
        """Draws the NEXT scenario of every environment that has none staged (mapf_stage_next) on the current stream; from then on
        `reset_envs` with the same (density, seed) hands the staged scenario over instead of drawing one.  Asynchronous."""
        check(lib.mapf_stage_next(self._h, float(density), int(seed) & 0xFFFFFFFFFFFFFFFF, _stream(self.device)), "mapf_stage_next")

    def set_agents(self, agents_pos, sync=True):
        """Overwrites the agent positions (rewind to the start of an action tape), steps := 0.  sync=False: asynchronous; the
        caller keeps `agents_pos` (an int16 device tensor, used as is) alive until the stream has consumed it."""
        a = agents_pos.to(self.device, torch.int16).contiguous()
        assert a.shape == (self.num_envs, self.num_agents, 2)
        check(lib.mapf_set_agents(self._h, _ptr(a), _stream(self.device)), "mapf_set_agents")
        if sync:
            torch.cuda.current_stream(self.device).synchronize()

    # -- the hot path --
    def step(self, actions, obs_out=None, obs_bits_out=None):
        """actions: int8 tensor [E,N] on this device.  Returns (obs, pos, reward, done, reward_class) --
        persistent buffers, overwritten by the next call (asynchronous on the current stream).
        obs_bits_out: optional int32/uint32 tensor [E, obs_bits_row_dwords] receiving the bit-packed observation."""
        assert actions.dtype == torch.int8 and actions.is_contiguous() and actions.shape == (self.num_envs, self.num_agents)
        obs = self.obs if obs_out is None else obs_out
        check(lib.mapf_step(self._h, _ptr(actions), _ptr(obs), _ptr(obs_bits_out), _ptr(self.pos), _ptr(self.reward_class),
                            _ptr(self.reward), _ptr(self.done), _stream(self.device)), "mapf_step")
        return obs, self.pos, self.reward, self.done, self.reward_class

    def observe(self, obs_out=None, obs_bits_out=None, mask=None):
        """(obs, pos) of the current state.  `mask` (uint8 tensor [E], optional): only the flagged environments' rows are
        rewritten -- for callers whose buffers already hold the other environments' current observation (the actor loop after
        step + reset_envs(mask))."""
        obs = self.obs if obs_out is None else obs_out
        if mask is not None:
            assert mask.dtype == torch.uint8 and mask.is_contiguous() and mask.shape == (self.num_envs,)
            check(lib.mapf_observe_masked(self._h, _ptr(mask), _ptr(obs), _ptr(obs_bits_out), _ptr(self.pos), _stream(self.device)),
                  "mapf_observe_masked")
        else:
            check(lib.mapf_observe(self._h, _ptr(obs), _ptr(obs_bits_out), _ptr(self.pos), _stream(self.device)), "mapf_observe")
        return obs, self.pos

    def load_envs(self, env_ids, maps, agents_pos, goals_pos):
        """Re-load only the listed environments (numpy host arrays) and rebuild their navi fields."""
        ids = np.ascontiguousarray(env_ids, dtype=np.int32)
        n = len(ids)
        if n == 0:
            return
        L, N = self.map_length, self.num_agents
        m = np.ascontiguousarray(np.asarray(maps) != 0, dtype=np.int8)
        a = np.ascontiguousarray(agents_pos, dtype=np.int16)
        g = np.ascontiguousarray(goals_pos, dtype=np.int16)
        assert m.shape == (n, L, L) and a.shape == (n, N, 2) and g.shape == (n, N, 2)
        check(lib.mapf_load_envs(self._h, ids.ctypes.data, n, m.ctypes.data, a.ctypes.data, g.ctypes.data,
                                 _stream(self.device)), "mapf_load_envs")

    def check_status(self):
        """Synchronises and raises the reference's exception for any sticky device-side error."""
        st = lib.mapf_check_status(self._h, _stream(self.device))
        if st == _lib.ERR_ACTION:
            raise AssertionError("action index out of range")
        if st == _lib.ERR_OVERLAP:
            raise RuntimeError("unique")
        check(st, "mapf_check_status")

    # -- state read-back --
    def navi_map(self):
        E, N, L = self.num_envs, self.num_agents, self.map_length
        out = torch.empty((E, N, 4, L, L), dtype=torch.uint8, device=self.device)
        check(lib.mapf_get_navi(self._h, _ptr(out), _stream(self.device)), "mapf_get_navi")
        return out

    def agents_pos(self):
        out = torch.empty((self.num_envs, self.num_agents, 2), dtype=torch.int16, device=self.device)
        check(lib.mapf_get_agents(self._h, _ptr(out), _stream(self.device)), "mapf_get_agents")
        return out

    def goals_pos(self):
        out = torch.empty((self.num_envs, self.num_agents, 2), dtype=torch.int16, device=self.device)
        check(lib.mapf_get_goals(self._h, _ptr(out), _stream(self.device)), "mapf_get_goals")
        return out

    def maps(self):
        out = torch.empty((self.num_envs, self.map_length, self.map_length), dtype=torch.int8, device=self.device)
        check(lib.mapf_get_maps(self._h, _ptr(out), _stream(self.device)), "mapf_get_maps")
        return out

    def steps(self):
        out = torch.empty((self.num_envs,), dtype=torch.int32, device=self.device)
        check(lib.mapf_get_steps(self._h, _ptr(out), _stream(self.device)), "mapf_get_steps")
        return out


class MultiEnvironment:
    """Several `VecEnvironment`s of DIFFERENT shapes stepped, reset and re-observed by one launch each (include/mapf_env.h:
    mapf_multi_*).  The reference draws a (num_agents, map side) level per episode inside one actor (environment.py:148-151,
    worker.py:422-428); here every active level of the curriculum is a handle, and this set is what steps them together.

    envs: the handles (their persistent `obs` / `pos` / `reward_class` / `reward` / `done` buffers receive the outputs, as with
    `VecEnvironment.step`); actions: per handle the int8 [E, N] buffer the step reads (persistent: its contents change, its address
    does not); obs_bits: per handle an int32 [E, row dwords] buffer for the bit-packed observation (or None); masks: per handle the
    uint8 [E] flags `reset()` and `observe_masked()` act on; reset_seeds: per handle the base of its scenario stream.  Raises MapfError(ERR_UNSUPPORTED) for shapes outside the merged
    launch's limits (more than 25 agents, more than 16 handles)."""

    def __init__(self, envs, actions, obs_bits, masks, reset_seeds=None):
        n = len(envs)
        assert n >= 1 and len(actions) == n and len(obs_bits) == n and len(masks) == n
        self.envs, self.device = list(envs), envs[0].device
        for e, a, b, m in zip(envs, actions, obs_bits, masks):
            assert a.dtype == torch.int8 and a.is_contiguous() and a.shape == (e.num_envs, e.num_agents)
            assert m.dtype == torch.uint8 and m.is_contiguous() and m.shape == (e.num_envs,)
            assert b is None or (b.is_contiguous() and b.shape == (e.num_envs, e.obs_bits_row_dwords))
        self._keep = (list(actions), list(obs_bits), list(masks), [e.obs for e in envs])
        arr = lambda ts: (ctypes.c_void_p * n)(*[None if t is None else t.data_ptr() for t in ts])
        self._h = ctypes.c_void_p()
        check(lib.mapf_multi_create(n, (ctypes.c_void_p * n)(*[e._h.value for e in envs]), arr(actions), arr([e.obs for e in envs]), arr(obs_bits),
                                    arr([e.pos for e in envs]), arr([e.reward_class for e in envs]), arr([e.reward for e in envs]),
                                    arr([e.done for e in envs]), arr(masks),
                                    None if reset_seeds is None else (ctypes.c_uint64 * n)(*[int(v) & 0xFFFFFFFFFFFFFFFF for v in reset_seeds]),
                                    ctypes.byref(self._h)), "mapf_multi_create")
        self.num_workgroups = lib.mapf_multi_num_workgroups(self._h)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value and lib is not None:
            lib.mapf_multi_destroy(h)
            self._h = ctypes.c_void_p()

    def step(self):
        """Environment.step of every handle on the actions in their buffers: one launch."""
        check(lib.mapf_multi_step(self._h, _stream(self.device)), "mapf_multi_step")

    def reset(self, density=-1.0, tick=None):
        """On-device reset of the environments flagged in the handles' masks: one launch; handle i draws from the scenario stream
        reset_seeds[i] + tick.  tick: optional int64 [1] device tensor, the caller's iteration counter (read on the device, so a
        captured graph draws new scenarios at every replay)."""
        check(lib.mapf_multi_reset(self._h, float(density), _ptr(tick), _stream(self.device)), "mapf_multi_reset")

    def observe_masked(self):
        """Re-observation of the flagged environments into the handles' buffers: one launch."""
        check(lib.mapf_multi_observe_masked(self._h, _stream(self.device)), "mapf_multi_observe_masked")


class Environment:
    """Source-compatible single environment (reference environment.py:74-467).

    Same constructor / reset / load / step / observe signatures and return types; the work runs on
    the GPU through a 1-environment handle.  `render`/`close` (matplotlib GUI) are out of scope.
    """

    def __init__(self, adaptive=False, map_length=None, num_agents=None, obs_radius=None, reward_fn=None, device=None):
        import config  # the entry-point compatible config module at the repo root

        map_length = config.map_length if map_length is None else map_length
        num_agents = config.num_agents if num_agents is None else num_agents
        obs_radius = config.obs_radius if obs_radius is None else obs_radius
        reward_fn = config.reward_fn if reward_fn is None else reward_fn
        self.adaptive = adaptive
        if adaptive:  # environment.py:92-94
            self.num_agents = config.init_set[0]
            self.map_size = (config.init_set[1], config.init_set[1])
        else:
            self.num_agents = num_agents
            self.map_size = (map_length, map_length)
        self.obs_radius = obs_radius
        self.reward_fn = reward_fn
        self._device = device
        self._vec = None
        self._seed_counter = 0
        self._generate()
        self.steps = 0

    # -- internals --
    def _ensure_vec(self):
        L, N = self.map_size[0], self.num_agents
        v = self._vec
        if v is None or v.map_length != L or v.num_agents != N:
            self._vec = VecEnvironment(1, L, N, self.obs_radius, self.reward_fn, self._device)
        return self._vec

    def _generate(self):
        # the reference draws from the global `random` / `numpy.random` state (environment.py:100-138);
        # here one 64-bit seed is drawn from `random` so that random.seed(k) makes runs reproducible
        seed = random.getrandbits(63)
        maps, agents, goals, _ = generate_scenarios(1, self.map_size[0], self.num_agents, -1.0, seed)
        self._install(maps[0], agents[0], goals[0])

    def _install(self, map_, agents_pos, goals_pos):
        self.map = np.copy(map_)
        self.agents_pos = np.array(agents_pos, dtype=np.int64)
        self.goals_pos = np.array(goals_pos, dtype=np.int64)
        self.num_agents = self.agents_pos.shape[0]
        self.map_size = (self.map.shape[0], self.map.shape[1])
        v = self._ensure_vec()
        v.load(np.asarray(self.map)[None], self.agents_pos[None], self.goals_pos[None])
        self._navi = None

    @property
    def navi_map(self):
        """bool [N,4,L+2r,L+2r], zero padded like reference environment.py:276."""
        if self._navi is None:
            r = self.obs_radius
            nv = self._vec.navi_map()[0].cpu().numpy().astype(bool)
            self._navi = np.pad(nv, ((0, 0), (0, 0), (r, r), (r, r)))
        return self._navi

    # -- reference API --
    def reset(self, level=None, num_agents=None, map_length=None):
        if self.adaptive:  # environment.py:148-151
            rand = random.choice(level)
            self.num_agents = rand[0]
            self.map_size = (rand[1], rand[1])
        elif num_agents is not None:
            self.num_agents = num_agents
            self.map_size = (map_length, map_length)
        self._generate()
        self.steps = 0
        return self.observe()

    def load(self, map: np.ndarray, agents_pos: np.ndarray, goals_pos: np.ndarray):
        """load map, use for testing (environment.py:198-215)"""
        self._install(map, agents_pos, goals_pos)
        self.steps = 0
        self.imgs = []

    def step(self, actions: List[int]):
        # environment.py:289-290
        assert len(actions) == self.num_agents, "actions number" + str(actions)
        assert all([action_idx < 5 and action_idx >= 0 for action_idx in actions]), "action index out of range"
        v = self._vec
        act = torch.tensor([list(actions)], dtype=torch.int8).to(v.device)
        obs, pos, _, done, rclass = v.step(act)
        v.check_status()
        rc = rclass[0].cpu().numpy()
        rewards = [self.reward_fn[REWARD_KEYS[k]] for k in rc]
        self.agents_pos = pos[0].cpu().numpy().astype(np.int64)
        self.steps += 1
        info = {"step": self.steps - 1}
        return (obs[0].cpu().numpy().astype(bool), self.agents_pos), rewards, bool(done[0].item()), info

    def observe(self):
        v = self._vec
        obs, pos = v.observe()
        self.agents_pos = pos[0].cpu().numpy().astype(np.int64)
        return obs[0].cpu().numpy().astype(bool), self.agents_pos

    def render(self):
        raise NotImplementedError("matplotlib rendering is out of scope (GUI only; reference environment.py:469-508)")

    def close(self, save=False):
        pass
