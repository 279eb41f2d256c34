"""CPU: pins oracle/replay_oracle.py (numpy restatement of buffer.py / worker.py replay logic) against
golden vectors captured from the unmodified reference."""
import numpy as np

from oracle import replay_oracle as RO
from tests import helpers as H
from tests import replay_golden as RG


def test_sumtree_known_answer():
    z = H.load_npz("dqn_replay.npz")
    st = RO.SumTree(8)
    st.batch_update(np.arange(8), np.arange(1, 9, dtype=np.float64))
    assert st.tree.tolist() == [36, 10, 26, 3, 7, 11, 15, 1, 2, 3, 4, 5, 6, 7, 8]
    assert np.array_equal(st.tree, z["st8_tree"])


def test_sumtree_sampling_and_update():
    z = H.load_npz("dqn_replay.npz")
    st = RO.SumTree(1024)
    st.batch_update(np.arange(1024), z["st1024_pri"])
    assert st.tree[0] == float(z["st1024_tree_root"])
    for k in range(3):
        idx, p = st.batch_sample(z["st1024_s%d_u" % k])
        assert np.array_equal(idx, z["st1024_s%d_idx" % k])
        assert np.array_equal(p, z["st1024_s%d_p" % k])
        assert np.all(p > 0)
    st.batch_update(z["st1024_upd_idx"], z["st1024_upd_p"])
    assert np.array_equal(st.tree, z["st1024_tree_after"])


def test_local_finish_priorities():
    z = H.load_npz("dqn_replay.npz")
    # td of the golden episodes was computed by the reference's LocalBuffer.finish from q/act/rew that the
    # generator did not store; check the formula on a hand case instead (buffer.py:173-177)
    q = np.array([[1, 2, 3, 0, 0], [0, 5, 1, 0, 0], [2, 2, 2, 2, 9]], np.float32)
    act = np.array([0, 1, 2], np.uint8)
    rew = np.array([-0.5, 3.0, -0.075], np.float16)
    td = RO.local_finish(q, act, rew, 3)
    r = rew.astype(np.float64)
    exp = np.abs(np.array([r[0] + 0.99 * r[1] + 3 - 1, r[1] + 0.99 * r[2] + 5 - 5, r[2] + 0 + 9 - 2]))
    assert np.allclose(td[:3], exp, rtol=0, atol=1e-12) and np.all(td[3:] == 0) and td.shape == (256,)


class _OracleAdapter:
    def __init__(self):
        self.gb = RO.GlobalBuffer(4)

    def add(self, ep):
        self.gb.add(ep["num_agents"], ep["obs"], ep["act"], ep["rew"], ep["hid"], ep["td"], ep["done"], ep["size"], ep["comm"])

    def sample(self, u):
        return self.gb.sample(u)

    def update_priorities(self, idx, p, old_ptr):
        self.gb.update_priorities(idx, p, old_ptr)

    def leaves(self):
        return self.gb.tree.tree[-self.gb.tree.capacity:]

    def tree_root(self):
        return float(self.gb.tree.tree[0])

    size = property(lambda self: self.gb.size)
    ptr = property(lambda self: self.gb.ptr)


def test_global_buffer_scenario():
    RG.run(H.load_npz("dqn_replay.npz"), _OracleAdapter())
