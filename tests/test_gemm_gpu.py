"""GPU: csrc/mapf_gemm.hip -- the learner's own dense products and reductions (reference worker.py:312-324: what loss.backward()
derives for the Linear / GRUCell weights) against plain PyTorch fp32 statements of the same operations."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _tall_ref(a, b):
    return a.float().t() @ b.float()


@pytest.mark.parametrize("K,m,n,dt", [
    (5000, 768, 256, torch.bfloat16),      # recurrent.weight_hh
    (26624, 384, 256, torch.bfloat16),     # W_Q | W_K | W_V
    (333, 64, 128, torch.bfloat16),        # K not a multiple of the 64-row block; W_O
    (4096, 768, 64, torch.bfloat16),       # update_cell.weight_ih
    (3000, 768, 784, torch.bfloat16),      # recurrent.weight_ih: the last column slab holds 16 of 64 columns
    (100000, 16, 128, torch.float16),      # the encoder's 1x1 head: a quarter of a row slab
    (70, 8, 8, torch.bfloat16), (1, 8, 16, torch.bfloat16), (64, 64, 64, torch.float16),
])
def test_tall_tn_against_fp32(K, m, n, dt):
    from mapf_rl_amd.fused import tall_tn_into

    g = torch.Generator(device="cuda").manual_seed(K + m)
    a = (torch.randn((K, m), device="cuda", generator=g) * 0.5).to(dt)
    b = (torch.randn((K, n), device="cuda", generator=g) * 0.5).to(dt)
    ref = _tall_ref(a, b)
    out = torch.full((m, n), 7.0, device="cuda")
    tall_tn_into(out, a, b)
    tol = 1e-4 * float(ref.abs().max()) + 1e-5
    assert float((out - ref).abs().max()) <= tol, float((out - ref).abs().max())
    # deterministic: the partial slabs are added in partition order
    out2 = torch.empty_like(out)
    tall_tn_into(out2, a, b)
    assert torch.equal(out, out2)
    # accumulate + scale (the encoder chain's 1 / loss scale travels as float bits at [1])
    scale = torch.tensor([0, np.float32(0.25).view(np.int32)], dtype=torch.int32, device="cuda")
    acc = torch.full((m, n), 3.0, device="cuda")
    tall_tn_into(acc, a, b, scale=scale, accumulate=True)
    assert float((acc - (3.0 + 0.25 * ref)).abs().max()) <= tol


def test_tall_tn_reads_strided_rows():
    from mapf_rl_amd.fused import tall_tn_into

    g = torch.Generator(device="cuda").manual_seed(3)
    wide = (torch.randn((9000, 1024), device="cuda", generator=g) * 0.3).to(torch.bfloat16)
    a, b = wide[:, 256:640], wide[:, 768:1024]           # column windows of a wider matrix: lda = ldb = 1024
    out = torch.empty((384, 256), device="cuda")
    for _ in range(3):
        tall_tn_into(out, a, b)
    assert float((out - _tall_ref(a, b)).abs().max()) <= 1e-4 * float(_tall_ref(a, b).abs().max())


def test_sum_parts_and_small_encoder_grads():
    from mapf_rl_amd._lib import check, lib
    from mapf_rl_amd.fused import sum_parts_into

    g = torch.Generator(device="cuda").manual_seed(5)
    parts = [torch.randn((128, 128, 3, 3, 128), device="cuda", generator=g) for _ in range(3)]
    outs = [torch.empty((128, 3, 3, 128), device="cuda") for _ in range(3)]
    scale = torch.tensor([0, np.float32(0.5).view(np.int32)], dtype=torch.int32, device="cuda")
    sum_parts_into(outs, parts, scale)
    for o, p in zip(outs, parts):
        assert torch.allclose(o, 0.5 * p.sum(0), rtol=1e-5, atol=1e-4)
    for nblk, rows7, P0 in ((37, 148, 512), (6001, 24004, 512), (3, 12, 7), (64, 263, 32)):
        _small_grads_case(nblk, rows7, P0, g)


def _small_grads_case(nblk, rows7, P0, g):
    from mapf_rl_amd._lib import check, lib

    gb, gb7 = torch.randn((7, nblk, 128), device="cuda", generator=g), torch.randn((rows7, 16), device="cuda", generator=g)
    ws0 = torch.randn((P0, 128, 64), device="cuda", generator=g)
    b7, b1, w0 = torch.empty((7, 128), device="cuda"), torch.empty(16, device="cuda"), torch.empty((128, 3, 3, 6), device="cuda")
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    scratch = torch.empty(65536, device="cuda")
    check(lib.mapf_encoder_small_grads(p(gb), nblk, p(b7), p(gb7), rows7, p(b1), p(ws0), P0, p(w0), p(scratch), st))
    assert torch.allclose(b7, gb.double().sum(1).float(), rtol=1e-5, atol=2e-3) and torch.allclose(b1, gb7.double().sum(0).float(), rtol=1e-5, atol=2e-3)
    ref0 = ws0.sum(0)[:, :54].view(128, 6, 3, 3).permute(0, 2, 3, 1)   # columns ci*9 + ky*3 + kx -> [co][ky][kx][ci]
    assert torch.allclose(w0, ref0, rtol=1e-5, atol=1e-3)


@pytest.mark.parametrize("rows", [1, 63, 64, 5000, 40000])
def test_latent_grad_rows_is_d_gi_times_w_ih(rows):
    from mapf_rl_amd._lib import check, lib
    from mapf_rl_amd.fused import LATGRAD_PACKED_ELEMS, latent_grad_rows

    g = torch.Generator(device="cuda").manual_seed(rows)
    w = torch.randn((768, 784), device="cuda", generator=g) * 0.05
    d = (torch.randn((rows, 768), device="cuda", generator=g) * 0.5).to(torch.bfloat16)
    packed = torch.empty(LATGRAD_PACKED_ELEMS, dtype=torch.bfloat16, device="cuda")
    check(lib.mapf_latent_grad_pack(ctypes.c_void_p(w.data_ptr()), ctypes.c_void_p(packed.data_ptr()), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
    out = latent_grad_rows(d, packed)
    ref = d.float() @ w.to(torch.bfloat16).float()
    assert out.shape == (rows, 784)
    assert float((out.float() - ref).abs().max()) <= 2e-2 * float(ref.abs().max())   # bf16 output rounding
