"""Autograd wrappers of the fused conv epilogues (include/mapf_dqn.h): y = relu(conv_out + bias (+ residual))
in one in-place pass over a bf16 NHWC activation, with a one-pass backward (masked gradient + bias-gradient
reduction).  Used by `Network.encode` on a HIP device under bf16 autocast."""
import ctypes

import torch

from ._lib import check, lib


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _is_nhwc(t):
    return t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last)


class _BiasResReLU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, bias, res):
        C = y.shape[1]
        check(lib.mapf_bias_res_relu_fwd(_ptr(y), _ptr(bias), _ptr(res), y.numel(), C, _stream(y.device)), "mapf_bias_res_relu_fwd")
        ctx.mark_dirty(y)
        ctx.save_for_backward(y)
        ctx.has_res = res is not None
        return y

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        C = y.shape[1]
        g = g.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        gx = torch.empty_like(y, memory_format=torch.channels_last)
        gb = torch.zeros(C, dtype=torch.float32, device=y.device)
        check(lib.mapf_bias_res_relu_bwd(_ptr(g), _ptr(y), _ptr(gx), _ptr(gb), y.numel(), C, _stream(y.device)), "mapf_bias_res_relu_bwd")
        return gx, gb, (gx if ctx.has_res else None)


def bias_res_relu(conv_out, bias, res=None):
    """conv_out: bf16 NHWC activation straight out of a bias-free convolution (modified in place);
    bias: f32 [C]; res: optional bf16 NHWC tensor of the same shape."""
    assert conv_out.dtype == torch.bfloat16 and _is_nhwc(conv_out) and bias.dtype == torch.float32
    if res is not None:
        assert res.dtype == torch.bfloat16 and _is_nhwc(res) and res.shape == conv_out.shape
    return _BiasResReLU.apply(conv_out, bias.contiguous(), res)
