"""On-device prioritized episode replay: host-side mirror of the reference's `GlobalBuffer`
(reference worker.py:21-250) and `SumTree` (reference buffer.py:16-105) on top of include/mapf_replay.h.

All episode data and the f64 sum tree live in HBM; `sample_batch` is a tree descent plus one gather kernel.
The ring pointer / size / counter live on the device too (a vectorised actor appends finished episodes without a host
round trip, `add_finished`); the curriculum statistics stay on the host like in the reference.
PyTorch is used for device memory and streams only."""
import ctypes
import threading

import numpy as np
import torch

from . import _lib
from ._lib import check, lib

MAX_STEPS, BT_STEPS, FORWARD_STEPS = 256, 16, 2
ALPHA, BETA = 0.6, 0.4  # reference config.py:42-43


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def pack_obs_rows(obs, max_agents, row_dwords):
    """bool/uint8 [R, na, 6, 9, 9] -> uint32 [R, row_dwords]: bit (a*486 + c*81 + cell), padded agents zero."""
    obs = np.asarray(obs)
    R, na = obs.shape[:2]
    full = np.zeros((R, max_agents, 486), dtype=np.uint8)
    full[:, :na] = obs.reshape(R, na, 486)
    bits = np.packbits(full.reshape(R, max_agents * 486), axis=1, bitorder="little")
    out = np.zeros((R, row_dwords * 4), dtype=np.uint8)
    out[:, :bits.shape[1]] = bits
    return out.view(np.uint32)


def pack_comm_rows(comm, max_agents):
    """bool [R, na, na] -> uint32 [R, A, CW]: bit j of word [a][j/32] = comm[r][a][j]."""
    comm = np.asarray(comm)
    R, na = comm.shape[:2]
    cw = (max_agents + 31) // 32
    full = np.zeros((R, max_agents, cw * 32), dtype=np.uint8)
    full[:, :na, :na] = comm
    return np.packbits(full, axis=2, bitorder="little").view(np.uint32).reshape(R, max_agents, cw)


class SumTree:
    """Device sum tree with the reference's interface (buffer.py:16-105), used through `GlobalBuffer` or alone."""

    def __init__(self, replay_handle, leaves, device):
        self._h, self.capacity, self.device = replay_handle, leaves, device

    def batch_update(self, idxes, priorities, alpha=0.0):
        idx = torch.as_tensor(idxes, dtype=torch.int64).to(self.device).contiguous()
        pri = torch.as_tensor(priorities, dtype=torch.float64).to(self.device).contiguous()
        check(lib.mapf_replay_tree_update(self._h, _ptr(idx), _ptr(pri), idx.numel(), float(alpha), _stream(self.device)),
              "mapf_replay_tree_update")
        torch.cuda.current_stream(self.device).synchronize()

    def batch_sample(self, batch_size, uniforms=None):
        u, unit = self._uniforms(batch_size, uniforms)
        idx = torch.empty(batch_size, dtype=torch.int64, device=self.device)
        pri = torch.empty(batch_size, dtype=torch.float64, device=self.device)
        check(lib.mapf_replay_tree_sample(self._h, _ptr(u), batch_size, unit, _ptr(idx), _ptr(pri), None, _stream(self.device)),
              "mapf_replay_tree_sample")
        return idx, pri

    def _uniforms(self, batch_size, uniforms):
        """(draws, unit flag): caller-supplied draws are already scaled to [0, sum/n) like np.random.uniform(0, interval)
        (buffer.py:60); own draws are U(0,1) and the kernel scales them by sum/n -- no copy of the root, no host round trip."""
        if uniforms is not None:
            return torch.as_tensor(uniforms, dtype=torch.float64).to(self.device).contiguous(), 0
        return torch.rand(batch_size, dtype=torch.float64, device=self.device), 1

    def tree(self):
        out = torch.empty(2 * self.capacity - 1, dtype=torch.float64, device=self.device)
        check(lib.mapf_replay_tree_read(self._h, _ptr(out), _stream(self.device)), "mapf_replay_tree_read")
        return out

    def sum(self):
        return float(self.tree()[0].item())  # diagnostic / tests only: copies the tree


class GlobalBuffer:
    """reference worker.py:21-250 (the Ray plumbing `run/prepare_data/get_data` is replaced by same-device calls)."""

    def __init__(self, capacity, max_agents=6, alpha=ALPHA, beta=BETA, device=None, init_set=(1, 10),
                 max_map_length=40, pass_rate=0.9, fixed_level=False):
        if not torch.cuda.is_available():
            raise RuntimeError("mapf_rl_amd.GlobalBuffer needs a HIP device (no CPU fallback)")
        assert abs(alpha - ALPHA) < 1e-12, "the priority exponent 0.6 is compiled into the add/update kernels"
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.capacity, self.max_agents, self.alpha, self.beta = capacity, max_agents, alpha, beta
        self._h = ctypes.c_void_p()
        check(lib.mapf_replay_create(capacity, max_agents, self.device.index, ctypes.byref(self._h)), "mapf_replay_create")
        self.row_dwords = lib.mapf_replay_row_dwords(self._h)
        self.priority_tree = SumTree(self._h, capacity * MAX_STEPS, self.device)
        self.lock = threading.Lock()
        # curriculum statistics (worker.py:32,74-82,205-250)
        from .curriculum import LevelTable

        self.init_set, self.max_map_length, self.pass_rate = tuple(init_set), max_map_length, pass_rate
        self.levels = LevelTable(init_set, max_agents, max_map_length, pass_rate, fixed=fixed_level)
        self._actors = []  # weak references to the vectorised actors recording into this buffer (their outcome logs)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value and lib is not None:
            lib.mapf_replay_destroy(h)
            self._h = ctypes.c_void_p()

    def state(self):
        """(ptr, size, counter, episodes added by the last add): the ring state lives on the device (the actor appends finished
        episodes without the host, `add_finished`); reading it waits for the current stream."""
        out = (ctypes.c_int64 * 4)()
        check(lib.mapf_replay_state(self._h, out, _stream(self.device)), "mapf_replay_state")
        return tuple(int(v) for v in out)

    def __len__(self):
        return self.state()[1]

    size = property(lambda self: len(self))
    ptr = property(lambda self: self.state()[0])
    counter = property(lambda self: self.state()[2])

    # ------------------------------------------------------------------ add
    def add_episode_device(self, num_agents, size, done, obs_bits, comm_bits, act, rew, hid, td):
        """All tensors already on the device in storage layout (see include/mapf_replay.h)."""
        assert td.numel() == MAX_STEPS and td.dtype == torch.float64 and td.is_contiguous()
        with self.lock:
            check(lib.mapf_replay_add(self._h, int(num_agents), int(size), int(bool(done)), _ptr(obs_bits), _ptr(comm_bits),
                                      _ptr(act), _ptr(rew), _ptr(hid), _ptr(td), _stream(self.device)), "mapf_replay_add")

    def add_episode(self, num_agents, obs, act, rew, hid, td_errors, done, size, comm):
        """One episode in the reference's array formats (LocalBuffer.finish tuple fields, buffer.py:179)."""
        d = self.device
        hid = np.asarray(hid)
        if hid.ndim == 3:
            hid = hid[:, 0]  # every agent row holds agent 0's state (quirk Q4)
        t = dict(
            obs=torch.from_numpy(pack_obs_rows(np.asarray(obs)[:size + 1], self.max_agents, self.row_dwords).view(np.int32)).to(d),
            comm=torch.from_numpy(pack_comm_rows(np.asarray(comm)[:size + 1], self.max_agents).view(np.int32)).to(d),
            act=torch.from_numpy(np.ascontiguousarray(np.asarray(act)[:size], np.uint8)).to(d),
            rew=torch.from_numpy(np.ascontiguousarray(np.asarray(rew)[:size], np.float16)).to(d),
            hid=torch.from_numpy(np.ascontiguousarray(hid[:size], np.float16)).to(d),
            td=torch.from_numpy(np.ascontiguousarray(td_errors, np.float64)).to(d))
        assert t["td"].numel() == MAX_STEPS
        self.add_episode_device(num_agents, size, done, t["obs"], t["comm"], t["act"], t["rew"], t["hid"], t["td"])
        torch.cuda.current_stream(d).synchronize()  # the staging tensors above are temporaries

    def add_finished(self, num_agents, finished, sizes, done, obs_bits, comm_bits, act, rew, hid, q):
        """GlobalBuffer.add (worker.py:71-104) + LocalBuffer.finish (buffer.py:153-179) for every environment flagged in
        `finished` (bool/u8 [E]) of a vectorised actor, in environment order, without any host read: see
        include/mapf_replay.h mapf_replay_add_many.  Local buffers: obs_bits int32 [E, S+1, RD], comm_bits int32 [E, S+1, A, CW],
        act u8 [E, S], rew f16 [E, S], hid f16 [E, S, 256], q f32 [E, S, 5]; sizes int64 [E]; done u8/bool [E]."""
        E, S = act.shape
        assert obs_bits.shape == (E, S + 1, self.row_dwords) and comm_bits.shape[:2] == (E, S + 1) and q.shape == (E, S, 5)
        assert sizes.dtype == torch.int64 and hid.shape == (E, S, 256)
        fin = finished.view(torch.uint8) if finished.dtype == torch.bool else finished
        dn = done.view(torch.uint8) if done.dtype == torch.bool else done
        assert fin.dtype == torch.uint8 and dn.dtype == torch.uint8
        with self.lock:
            check(lib.mapf_replay_add_many(self._h, E, int(num_agents), S, _ptr(fin), _ptr(sizes), _ptr(dn), _ptr(obs_bits), _ptr(comm_bits),
                                           _ptr(act), _ptr(rew), _ptr(hid), _ptr(q), _stream(self.device)), "mapf_replay_add_many")

    def add_finished_env(self, num_agents_env, finished, sizes, done, obs_bits, comm_bits, act, rew, hid, q):
        """`add_finished` for environments of different agent counts (several curriculum levels, their local buffers back to back):
        num_agents_env int32 [E] (include/mapf_replay.h: mapf_replay_add_many_env)."""
        E, S = act.shape
        assert obs_bits.shape == (E, S + 1, self.row_dwords) and comm_bits.shape[:2] == (E, S + 1) and q.shape == (E, S, 5)
        assert sizes.dtype == torch.int64 and hid.shape == (E, S, 256) and num_agents_env.dtype == torch.int32 and num_agents_env.shape == (E,)
        assert finished.dtype == torch.uint8 and done.dtype == torch.uint8
        with self.lock:
            check(lib.mapf_replay_add_many_env(self._h, E, _ptr(num_agents_env), S, _ptr(finished), _ptr(sizes), _ptr(done), _ptr(obs_bits),
                                               _ptr(comm_bits), _ptr(act), _ptr(rew), _ptr(hid), _ptr(q), _stream(self.device)), "mapf_replay_add_many_env")

    def add(self, buffer_list):
        """reference signature (worker.py:71): list of LocalBuffer.finish() tuples
        (actor_id, num_agents, map_len, obs, act, rew, hid, td_errors, done, size, comm_mask)."""
        for b in buffer_list:
            if b[0] >= 10:  # curriculum statistics only from actors with id >= 10 (worker.py:74)
                self.levels.record((b[1], b[2]), b[8])
        for b in buffer_list:
            self.add_episode(b[1], b[3], b[4], b[5], b[6], b[7], b[8], b[9], b[10])

    # ------------------------------------------------------------------ sample
    def make_batch_slot(self, batch_size):
        """Persistent output buffers for `sample_batch(..., out=slot)`: every sample lands at the same addresses (what a captured HIP
        graph of the learner's update needs, update.FusedUpdate)."""
        d, A, B = self.device, self.max_agents, batch_size
        return dict(B=B, A=A, old_ptr=torch.empty(1, dtype=torch.int64, device=d), idx=torch.empty(B, dtype=torch.int64, device=d),
                    pri=torch.empty(B, dtype=torch.float64, device=d), obs=torch.empty((18, B, A, 6, 9, 9), dtype=torch.bfloat16, device=d),
                    comm=torch.empty((18, B, A, A), dtype=torch.uint8, device=d), hidden=torch.empty((B * A, 256), dtype=torch.float16, device=d),
                    action=torch.empty(B, dtype=torch.int64, device=d), reward=torch.empty(B, dtype=torch.float32, device=d),
                    done=torch.empty(B, dtype=torch.float32, device=d), steps=torch.empty(B, dtype=torch.float32, device=d),
                    bt=torch.empty(B, dtype=torch.int64, device=d), weights=torch.empty(B, dtype=torch.float32, device=d))

    def sample_batch(self, batch_size, uniforms=None, out=None):
        """Returns the reference's 11-tuple (worker.py:168-182) as device tensors:
        (obs bf16 [B,18,A,6,9,9], action i64 [B,1], reward f32 [B,1], done f32 [B,1], steps f32 [B,1],
         bt_steps i64 [B], hidden f16 [B*A,256], comm_mask bool [B,18,A,A], idxes i64 [B], weights f32 [B,1],
         old_ptr 0-dim i64 device tensor).  out: a `make_batch_slot` dict to write into (the tuple then consists of views of it)."""
        d, A, B = self.device, self.max_agents, batch_size
        if out is None:
            out = self.make_batch_slot(B)
        assert out["B"] == B and out["A"] == A
        with self.lock:
            u, unit = self.priority_tree._uniforms(B, uniforms)
            # time-major in memory (the learner's recurrence wants [T, B, ...] and would otherwise transpose 130 MB per network);
            # handed out as [B, 18, ...] views: the reference's shape
            old_ptr, idx, pri, obs, comm, hidden = out["old_ptr"], out["idx"], out["pri"], out["obs"], out["comm"], out["hidden"]
            action, reward, done, steps, bt = out["action"], out["reward"], out["done"], out["steps"], out["bt"]
            check(lib.mapf_replay_sample(self._h, _ptr(u), B, unit, _ptr(idx), _ptr(pri), _ptr(obs), _ptr(comm), _ptr(hidden),
                                         _ptr(action), _ptr(reward), _ptr(done), _ptr(steps), _ptr(bt), _ptr(old_ptr), _stream(d)),
                  "mapf_replay_sample")
            old_ptr = old_ptr[0]  # 0-dim device tensor: the ring pointer at sample time (worker.py:182); int(old_ptr) reads it
        check(lib.mapf_replay_is_weights(_ptr(pri), B, float(self.beta), _ptr(out["weights"]), _stream(d)), "mapf_replay_is_weights")  # worker.py:165-166
        # (the gather kernel writes 0 / 1 bytes: the bool tensor is a view)
        return (obs.transpose(0, 1), action.unsqueeze(1), reward.unsqueeze(1), done.unsqueeze(1), steps.unsqueeze(1), bt, hidden,
                comm.view(torch.bool).transpose(0, 1), idx, out["weights"].unsqueeze(1), old_ptr)

    def update_priorities(self, idxes, priorities, old_ptr):
        """worker.py:186-203; idxes / priorities: device tensors (or array-likes)."""
        idx = torch.as_tensor(idxes, dtype=torch.int64).to(self.device).contiguous()
        pri = torch.as_tensor(priorities).to(self.device, torch.float64).contiguous()
        old = torch.as_tensor(old_ptr, dtype=torch.int64).to(self.device).reshape(1).contiguous()  # int or the sample's tensor
        with self.lock:
            check(lib.mapf_replay_update_priorities(self._h, _ptr(idx), _ptr(pri), idx.numel(), _ptr(old), _stream(self.device)),
                  "mapf_replay_update_priorities")
        self._keep = (idx, pri, old)  # keep the (possibly temporary) tensors alive until the next call

    # ------------------------------------------------------------------ curriculum / stats (worker.py:205-250)
    # The schedule itself lives in curriculum.LevelTable; these are the reference's names for it.
    stat_dict = property(lambda self: self.levels.windows, lambda self, v: setattr(self.levels, "windows", {tuple(k): list(w) for k, w in v.items()}))
    level = property(lambda self: self.levels.levels)

    def register_actor(self, actor):
        import weakref

        self._actors.append(weakref.ref(actor))

    def drain_outcomes(self):
        """Episode outcomes the actors logged on the device -> the level table's windows (actor.VecActor.drain_outcomes)."""
        self._actors = [a for a in self._actors if a() is not None]
        for a in self._actors:
            a().drain_outcomes()

    def pooled_counts(self, device, group=None):
        """Multi-rank: every rank's per-level (successes, episodes), summed (curriculum.LevelTable.pooled_counts)."""
        self.drain_outcomes()
        return self.levels.pooled_counts(device, group)

    def stats(self, interval, pooled=None, world=1):
        """Prints the reference's lines (worker.py:206-210) and advances the curriculum; `pooled` = self.pooled_counts()
        in a multi-rank run (every rank must then call this with the same counts)."""
        self.drain_outcomes()
        print("buffer update speed: {}/s".format(int(lib.mapf_replay_counter(self._h, 1)) / interval))
        print("buffer size: {}".format(len(self)))
        for line in self.levels.advance(pooled, self.levels.WINDOW * world):
            print(line)

    def advance_levels(self, pooled=None, world=1):
        """The promotion rule alone (worker.py:211-224), without the statistics' lines: `train.py --promote-interval` checks it more
        often than it prints.  Returns the per-level lines `stats` would print."""
        self.drain_outcomes()
        return self.levels.advance(pooled, self.levels.WINDOW * world)

    def ready(self, learning_starts=50000):
        return len(self) >= learning_starts

    def get_level(self):
        return self.levels.levels

    def check_done(self, pooled=None, world=1):
        if pooled is None:
            self.drain_outcomes()
        return self.levels.done(pooled, self.levels.WINDOW * world)
