"""Shared driver: replays the scenario of tests/golden/dqn_replay.npz (episodes added to a 4-slot
GlobalBuffer, samples after episodes 3 and 5, a fresh and a STALE priority update) against any buffer
implementation exposing add / sample / update_priorities / leaves, and checks every golden output."""
import numpy as np

from tests import helpers as H


def episodes(z):
    for k in range(int(z["gb_num_eps"])):
        pre = "gb_ep%d_" % k
        aid, na, ml, size, done = [int(v) for v in z[pre + "meta"]]
        obs = H.unpack_bits(z[pre + "obs_bits"], (size + 1, na, 6, 9, 9)).astype(bool)
        yield k, dict(actor_id=aid, num_agents=na, map_len=ml, size=size, done=bool(done), obs=obs,
                      act=z[pre + "act"], rew=z[pre + "rew"].astype(np.float16), hid=z[pre + "hid"].astype(np.float16),
                      td=z[pre + "td"], comm=z[pre + "comm"])


def check_sample(z, tag, out, A=6):
    """out: dict with numpy arrays obs [B,18,A,6,9,9] (any dtype, 0/1), action, reward, done, steps, bt_steps,
    hidden [B*A,256], comm_mask, idxes, weights, old_ptr."""
    B = len(z[tag + "idxes"])
    assert np.array_equal(np.asarray(out["idxes"]), z[tag + "idxes"]), tag
    ref_obs = H.unpack_bits(z[tag + "obs_bits"], (B, 18, A, 6, 9, 9))
    assert np.array_equal(np.asarray(out["obs"]).astype(np.uint8), ref_obs), tag
    assert np.array_equal(np.asarray(out["action"]).reshape(B), z[tag + "action"].reshape(B))
    assert np.array_equal(np.asarray(out["reward"], np.float32).reshape(B), z[tag + "reward"].reshape(B))
    assert np.array_equal(np.asarray(out["done"], np.float32).reshape(B), z[tag + "done"].reshape(B))
    assert np.array_equal(np.asarray(out["steps"], np.float32).reshape(B), z[tag + "steps"].reshape(B))
    assert np.array_equal(np.asarray(out["bt_steps"]).reshape(B), z[tag + "bt_steps"])
    assert np.array_equal(np.asarray(out["hidden"], np.float32), z[tag + "hidden"]), tag
    assert np.array_equal(np.asarray(out["comm_mask"]).astype(bool), z[tag + "comm"]), tag
    w = np.asarray(out["weights"], np.float32).reshape(B)
    assert np.allclose(w, z[tag + "weights"].reshape(B), rtol=2e-3, atol=0), tag  # reference stores f16
    assert int(out["old_ptr"]) == int(z[tag + "old_ptr"])


def _exact(a, b):
    return np.array_equal(np.asarray(a), np.asarray(b))


def run(z, buf, leaves_equal=_exact, root_equal=_exact):
    """`leaves_equal` / `root_equal`: how sum-tree leaves (td ** 0.6) and the root are compared -- exactly for a numpy
    implementation, within an explicit relative bound for the device (its f64 pow() is <= 2 ulp from numpy's)."""
    stale = None
    for k, ep in episodes(z):
        buf.add(ep)
        if k in (3, 5):
            tag = "gb_s%d_" % k
            out = buf.sample(z[tag + "u"])
            check_sample(z, tag, out)
            assert buf.size == int(z[tag + "size"])
            assert root_equal(buf.tree_root(), float(z[tag + "tree_root"]))
            if k == 3:
                buf.update_priorities(np.asarray(out["idxes"]).copy(), z[tag + "newp"].copy(), int(out["old_ptr"]))
                assert leaves_equal(buf.leaves(), z[tag + "leaves_after"])
                stale = (np.asarray(out["idxes"]).copy(), int(out["old_ptr"]))
    buf.update_priorities(stale[0], z["gb_stale_newp"].copy(), stale[1])
    assert leaves_equal(buf.leaves(), z["gb_stale_leaves_after"])
    assert buf.ptr == int(z["gb_final_ptr"])
