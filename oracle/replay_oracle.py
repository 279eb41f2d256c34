"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the reference's replay path, used to check the device
replay (mapf_rl_amd/replay.py + csrc/mapf_replay.hip).  Never imported by the product.

Parity status: PINNED against tests/golden/dqn_replay.npz (captured from the unmodified reference by
tests/golden/make_dqn_goldens.py) in tests/test_replay_oracle.py.

Follows, line by line:
  SumTree                 reference buffer.py:16-105
  local_finish            reference buffer.py:153-179 (LocalBuffer.finish: initial priorities)
  GlobalBuffer.add        reference worker.py:71-104
  GlobalBuffer.sample     reference worker.py:106-184
  GlobalBuffer.update_priorities   reference worker.py:186-203
"""
import numpy as np

MAX_STEPS = 256      # config.py:29 (= local_buffer_size, config.py:33)
BT_STEPS = 16        # config.py:30
FORWARD_STEPS = 2    # config.py:65
ALPHA, BETA = 0.6, 0.4  # config.py:42-43


class SumTree:
    def __init__(self, capacity):
        layer = 1
        while 2 ** (layer - 1) < capacity:
            layer += 1
        assert 2 ** (layer - 1) == capacity, "buffer size only support power of 2 size"  # buffer.py:23
        self.layer, self.capacity = layer, capacity
        self.tree = np.zeros(2 ** layer - 1, dtype=np.float64)

    def batch_update(self, idxes, priorities):
        idxes = np.asarray(idxes, dtype=np.int64) + self.capacity - 1  # buffer.py:96 (the reference mutates the caller's array)
        self.tree[idxes] = priorities
        for _ in range(self.layer - 1):
            idxes = np.unique((idxes - 1) // 2)
            self.tree[idxes] = self.tree[2 * idxes + 1] + self.tree[2 * idxes + 2]

    def prefixsums(self, uniforms):
        """buffer.py:57-62 with the uniform draws injected: arange(0, sum, interval)[k] = k * interval."""
        n = len(uniforms)
        interval = self.tree[0] / n
        ps = np.arange(n, dtype=np.float64) * interval + np.asarray(uniforms, dtype=np.float64)
        if ps[0] == 0:
            ps[0] = 1e-5
        return ps

    def batch_sample(self, uniforms):
        ps = self.prefixsums(uniforms)
        idxes = np.zeros(len(ps), dtype=np.int64)
        for _ in range(self.layer - 1):  # buffer.py:66-70
            p = self.tree[idxes * 2 + 1]
            idxes = np.where(ps <= p, idxes * 2 + 1, idxes * 2 + 2)
            ps = np.where(idxes % 2 == 0, ps - self.tree[idxes - 1], ps)
            ps = np.where(ps == 0, 1e-5, ps)
        priorities = self.tree[idxes]
        return idxes - (self.capacity - 1), priorities


def local_finish(q_buf, act_buf, rew_buf, size, capacity=MAX_STEPS):
    """Initial priorities of one episode (buffer.py:170-177).  q_buf f32 [size(+1), 5], act u8 [size], rew f16 [size]."""
    td = np.zeros(capacity, dtype=np.float64)
    q_max = np.max(q_buf[:size], axis=1)
    ret = rew_buf[:size].tolist() + [0 for _ in range(FORWARD_STEPS - 1)]
    reward = np.convolve(ret, [0.99 ** (FORWARD_STEPS - 1 - i) for i in range(FORWARD_STEPS)], "valid") + q_max
    q_val = q_buf[np.arange(size), act_buf[:size]]
    td[:size] = np.abs(reward - q_val)
    return td


class GlobalBuffer:
    """Ring of `capacity` episode slots; agent dimension padded to `max_agents` (config.max_num_agetns = 6)."""

    def __init__(self, capacity, max_agents=6, latent=256):
        self.capacity, self.max_agents = capacity, max_agents
        self.size = self.ptr = self.counter = 0
        self.tree = SumTree(capacity * MAX_STEPS)
        A = max_agents
        self.obs_buf = np.zeros(((MAX_STEPS + 1) * capacity, A, 6, 9, 9), dtype=bool)
        self.act_buf = np.zeros(MAX_STEPS * capacity, dtype=np.uint8)
        self.rew_buf = np.zeros(MAX_STEPS * capacity, dtype=np.float16)
        self.hid_buf = np.zeros((MAX_STEPS * capacity, A, latent), dtype=np.float16)
        self.done_buf = np.zeros(capacity, dtype=bool)
        self.size_buf = np.zeros(capacity, dtype=np.uint64)
        self.comm_mask = np.zeros(((MAX_STEPS + 1) * capacity, A, A), dtype=bool)

    def add(self, num_agents, obs, act, rew, hid, td_errors, done, size, comm, zero_padding=False):
        """worker.py:86-104.  obs bool [size+1, na, 6,9,9]; hid [size, na, 256] (or [size, 256], broadcast: quirk Q4).
        zero_padding=True clears the agent rows >= num_agents of the slot first (the product's declared
        deviation, include/mapf_replay.h); the reference leaves the previous episode's data there."""
        p = self.ptr
        if zero_padding:
            self.obs_buf[p * (MAX_STEPS + 1):(p + 1) * (MAX_STEPS + 1)] = False
            self.comm_mask[p * (MAX_STEPS + 1):(p + 1) * (MAX_STEPS + 1)] = False
            self.hid_buf[p * MAX_STEPS:(p + 1) * MAX_STEPS] = 0
        idxes = np.arange(p * MAX_STEPS, (p + 1) * MAX_STEPS)
        start = p * MAX_STEPS
        self.size -= int(self.size_buf[p])
        self.size += size
        self.counter += size
        self.tree.batch_update(idxes, np.asarray(td_errors, np.float64) ** ALPHA)
        self.obs_buf[start + p:start + p + size + 1, :num_agents] = obs
        self.act_buf[start:start + size] = act
        self.rew_buf[start:start + size] = rew
        self.hid_buf[start:start + size, :num_agents] = hid
        self.done_buf[p] = done
        self.size_buf[p] = size
        self.comm_mask[start + p:start + p + size + 1, :num_agents, :num_agents] = comm
        self.ptr = (p + 1) % self.capacity

    def sample(self, uniforms):
        """worker.py:114-182 with the uniforms injected.  Returns a dict of numpy arrays (reference dtypes)."""
        B = len(uniforms)
        A = self.max_agents
        idxes, priorities = self.tree.batch_sample(uniforms)
        T = BT_STEPS + FORWARD_STEPS
        out = dict(obs=np.zeros((B, T, A, 6, 9, 9), bool), action=np.zeros((B, 1), np.int64),
                   reward=np.zeros((B, 1), np.float16), done=np.zeros((B, 1), np.float16),
                   steps=np.zeros((B, 1), np.float16), bt_steps=np.zeros(B, np.int64),
                   hidden=np.zeros((B * A, 256), np.float16), comm_mask=np.zeros((B, T, A, A), bool))
        for b, idx in enumerate(idxes):
            g, l = idx // MAX_STEPS, idx % MAX_STEPS
            assert l < self.size_buf[g]
            steps = int(min(FORWARD_STEPS, int(self.size_buf[g]) - l))
            first_row = g * (MAX_STEPS + 1)
            if l < BT_STEPS - 1:
                lo, hidden = first_row, None
                clo = first_row
            elif l == BT_STEPS - 1:
                lo, hidden = idx + g + 1 - BT_STEPS, None
                clo = first_row  # worker.py:131 (same row as lo when l == 15)
            else:
                lo = clo = idx + g + 1 - BT_STEPS
                hidden = self.hid_buf[idx - BT_STEPS]
            hi = idx + g + 1 + steps
            n = hi - lo
            out["obs"][b, :n] = self.obs_buf[lo:hi]
            out["comm_mask"][b, :hi - clo] = self.comm_mask[clo:hi]
            if hidden is not None:
                out["hidden"][b * A:(b + 1) * A] = hidden
            out["action"][b, 0] = self.act_buf[idx]
            out["reward"][b, 0] = self.rew_buf[idx]
            out["done"][b, 0] = bool(l == self.size_buf[g] - 1 and self.done_buf[g])
            out["steps"][b, 0] = steps
            out["bt_steps"][b] = min(l + 1, BT_STEPS)
        min_p = np.min(priorities)
        out["weights"] = np.power(priorities / min_p, -BETA).astype(np.float16)[:, None]
        out["idxes"] = idxes
        out["priorities"] = priorities
        out["old_ptr"] = self.ptr
        return out

    def update_priorities(self, idxes, priorities, old_ptr):
        """worker.py:186-203: drop samples whose slot was overwritten since `old_ptr`."""
        idxes = np.asarray(idxes, np.int64)
        priorities = np.asarray(priorities, np.float64)
        if self.ptr > old_ptr:
            mask = (idxes < old_ptr * MAX_STEPS) | (idxes >= self.ptr * MAX_STEPS)
            idxes, priorities = idxes[mask], priorities[mask]
        elif self.ptr < old_ptr:
            mask = (idxes < old_ptr * MAX_STEPS) & (idxes >= self.ptr * MAX_STEPS)
            idxes, priorities = idxes[mask], priorities[mask]
        self.tree.batch_update(idxes, priorities ** ALPHA)
