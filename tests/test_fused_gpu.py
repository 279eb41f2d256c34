"""GPU: the hand-written conv epilogue kernels (include/mapf_dqn.h) against plain PyTorch fp32 math of the
same op, forward and backward, and the fused encoder against the unfused one (same bf16 convolutions)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("M,C,HW,with_res", [(37, 128, 7, True), (64, 128, 7, False), (5, 16, 7, False), (300, 128, 7, True)])
def test_bias_res_relu_forward_backward(M, C, HW, with_res):
    from mapf_rl_amd.fused import bias_res_relu

    torch.manual_seed(M + C)
    cl = torch.channels_last
    y0 = torch.randn(M, C, HW, HW, device="cuda").to(torch.bfloat16).contiguous(memory_format=cl)
    res0 = torch.randn(M, C, HW, HW, device="cuda").to(torch.bfloat16).contiguous(memory_format=cl) if with_res else None
    bias = torch.randn(C, device="cuda", requires_grad=True)
    g = torch.randn(M, C, HW, HW, device="cuda").to(torch.bfloat16).contiguous(memory_format=cl)
    # reference: fp32 math on the same bf16 inputs
    yr = y0.float().requires_grad_(True)
    rr = res0.float().requires_grad_(True) if with_res else None
    br = bias.detach().clone().requires_grad_(True)
    out_ref = torch.relu(yr + br.view(1, C, 1, 1) + (rr if with_res else 0))
    out_ref.backward(g.float())
    # fused (in place on a non-leaf, like a conv output)
    yl = y0.clone().requires_grad_(True)
    rl = res0.clone().requires_grad_(True) if with_res else None
    out = bias_res_relu(yl * 1, bias, rl * 1 if with_res else None)
    out.backward(g)
    assert torch.allclose(out.float(), out_ref, rtol=1e-2, atol=1e-2)  # bf16 rounding of the result
    mask = out_ref > 0
    assert torch.equal(out > 0, mask) or (out.float() - out_ref).abs().max() < 1e-2
    gx_ref = yr.grad
    assert torch.allclose(yl.grad.float(), gx_ref, rtol=0, atol=1e-6) or ((yl.grad.float() - gx_ref).abs() > 0).float().mean() < 1e-3
    if with_res:
        assert torch.equal(rl.grad, yl.grad)
    assert torch.allclose(bias.grad, br.grad, rtol=2e-2, atol=2e-2 * float(br.grad.abs().max()))


def test_fused_encoder_matches_unfused():
    from mapf_rl_amd.model import Network

    torch.manual_seed(0)
    net = Network().cuda()
    obs = (torch.rand(513, 6, 9, 9, device="cuda") < 0.3).to(torch.uint8)
    outs, grads = [], []
    Network.FUSED_TRAINING = False  # this test is about the epilogue kernels of the layer-by-layer path
    for fused in (True, False):
        Network.FUSED_EPILOGUE = fused
        net.zero_grad()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            lat = net.encode(obs)
            loss = (lat.float() ** 2).mean()
        loss.backward()
        outs.append(lat.float().detach().clone())
        grads.append({k: p.grad.detach().clone() for k, p in net.obs_encoder.named_parameters()})
    Network.FUSED_EPILOGUE = True
    Network.FUSED_TRAINING = True
    assert outs[0].shape == (513, 784)
    assert torch.allclose(outs[0], outs[1], rtol=3e-2, atol=3e-2)
    for k in grads[0]:
        a, b = grads[0][k], grads[1][k]
        assert torch.allclose(a, b, rtol=5e-2, atol=5e-2 * float(b.abs().max()) + 1e-8), k


@pytest.mark.parametrize("E,N,L,k", [(1, 1, 5, 3), (3, 2, 4, 3), (64, 40, 32, 3), (17, 128, 64, 3), (9, 33, 12, 5), (5, 6, 20, 1)])
def test_comm_mask_kernel_matches_torch_rule(E, N, L, k):
    """mapf_comm_mask against the torch formulation of model.py:195-208 (int64 path of comm_mask_from_pos), incl.
    distance ties (dense small maps) and the replay's packed row format."""
    from mapf_rl_amd.actor import pack_comm_device
    from mapf_rl_amd.fused import comm_mask
    from mapf_rl_amd.model import comm_mask_from_pos

    g = torch.Generator().manual_seed(E * 1000 + N)
    # distinct cells per environment, like real agent positions
    cells = torch.stack([torch.randperm(L * L, generator=g)[:N] for _ in range(E)])
    pos = torch.stack([cells // L, cells % L], dim=-1)
    ref = comm_mask_from_pos(pos.cuda().to(torch.int64), 4, k)          # torch path
    cw = (N + 31) // 32 + 1
    mask, packed = comm_mask(pos.cuda().to(torch.int16), 4, k, packed_words=cw)
    assert mask.dtype == torch.bool and torch.equal(mask, ref)
    assert torch.equal(comm_mask_from_pos(pos.cuda().to(torch.int16), 4, k), ref)  # dispatches to the kernel
    assert torch.equal(packed, pack_comm_device(ref, cw))
    assert bool(mask.diagonal(dim1=1, dim2=2).all())                      # an agent always hears itself
    assert int(mask.sum(-1).max()) <= min(k, N)
