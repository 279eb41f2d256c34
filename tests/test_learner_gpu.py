"""GPU (bf16 autocast): Learner.update against the reference golden (fp32): td error / loss within the bf16
tolerance, and the end-to-end device path replay.sample_batch -> update -> update_priorities."""
import numpy as np
import pytest
import torch

from tests import helpers as H
from tests.test_learner_cpu import _batch, _models

pytestmark = pytest.mark.gpu


def test_update_bf16_vs_reference():
    z = H.load_npz("dqn_update.npz")
    lr = _models("cuda")
    out = lr.update(_batch(z, "cuda", torch.bfloat16))
    td, ref = out["td"].float().cpu().numpy(), z["td"]
    assert np.all(np.abs(td - ref) <= 4e-2 * np.maximum(1.0, np.abs(ref))), np.abs(td - ref).max()  # 2 bootstraps: 2 x 2e-2
    assert abs(float(out["loss"]) - float(z["loss"])) <= 5e-2 * max(1.0, float(z["loss"]))
    assert abs(float(out["grad_norm"]) - float(z["grad_norm"])) <= 0.1 * float(z["grad_norm"])


def test_device_pipeline_sample_update_priorities():
    from mapf_rl_amd.learner import Learner
    from mapf_rl_amd.replay import GlobalBuffer

    torch.manual_seed(0)
    rng = np.random.RandomState(0)
    A = 4
    buf = GlobalBuffer(8, max_agents=A)
    for k in range(8):
        size = int(rng.randint(20, 60))
        td = np.zeros(256)
        td[:size] = rng.random_sample(size) + 0.1
        buf.add_episode(A, rng.random_sample((size + 1, A, 6, 9, 9)) < 0.3, rng.randint(0, 5, size).astype(np.uint8),
                        rng.choice([-0.075, -0.5, 3.0], size).astype(np.float16), (rng.standard_normal((size, 256)) * 0.3).astype(np.float16),
                        td, bool(k % 2), size, rng.random_sample((size + 1, A, A)) < 0.5)
    lr = Learner(buf, device="cuda", batch_size=32)
    before = buf.priority_tree.tree().clone()
    losses = [float(lr.update()["loss"]) for _ in range(3)]
    after = buf.priority_tree.tree()
    assert all(np.isfinite(losses)) and lr.counter == 3
    assert not torch.equal(before, after)            # priorities were written back
    leaves = after[-buf.priority_tree.capacity:]
    assert abs(float(after[0]) - float(leaves.sum())) < 1e-6 * float(after[0])   # root == sum of leaves (buffer.py:30)


def test_pipelined_target_forward_equals_sequential_order():
    """Learner(prefetch=True) samples batch k+1 and runs its target-network forward on a second stream while update k's backward
    is still in flight; the numbers must be those of the sequential order (same samples: the priorities of update k are
    written before batch k+1 is drawn; same target: nothing the backward does touches the target network)."""
    from mapf_rl_amd.learner import Learner
    from mapf_rl_amd.model import Network
    from mapf_rl_amd.replay import GlobalBuffer

    def run(prefetch):
        torch.manual_seed(0)
        rng = np.random.RandomState(1)
        A = 5
        buf = GlobalBuffer(8, max_agents=A)
        for k in range(8):
            size = int(rng.randint(30, 70))
            td = np.zeros(256)
            td[:size] = rng.random_sample(size) + 0.1
            buf.add_episode(A, rng.random_sample((size + 1, A, 6, 9, 9)) < 0.3, rng.randint(0, 5, size).astype(np.uint8),
                            rng.choice([-0.075, -0.5, 3.0], size).astype(np.float16), (rng.standard_normal((size, 256)) * 0.3).astype(np.float16),
                            td, bool(k % 2), size, rng.random_sample((size + 1, A, A)) < 0.5)
        torch.manual_seed(7)
        torch.cuda.manual_seed(7)
        lr = Learner(buf, device="cuda", batch_size=24, model=Network(), prefetch=prefetch)
        assert lr.prefetch == prefetch
        outs = [lr.update() for _ in range(4)]
        torch.cuda.synchronize()
        return outs, [p.detach().clone() for p in lr.model.parameters()], buf.priority_tree.tree().clone()

    (o_seq, p_seq, t_seq), (o_pre, p_pre, t_pre) = run(False), run(True)
    for a, b in zip(o_seq, o_pre):
        assert torch.equal(a["q_next"], b["q_next"]) and torch.equal(a["td"], b["td"])
        assert float(a["loss"]) == float(b["loss"])
    for a, b in zip(p_seq, p_pre):
        assert torch.equal(a, b)
    assert torch.equal(t_seq, t_pre)
