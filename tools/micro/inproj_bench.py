#!/usr/bin/env python3
"""mapf_input_proj_rows (csrc/mapf_inproj.hip) against the library GEMM: values (fp32 reference of the same bf16 operands) and time,
for all rows and for row lists of different lengths.  Run on the GPU."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mapf_rl_amd._lib import check, lib  # noqa: E402
from mapf_rl_amd.fused import INPROJ_PACKED_ELEMS, _ptr, input_proj_rows, mm_rows  # noqa: E402

dev = "cuda"
torch.manual_seed(0)
w = torch.randn(768, 784, device=dev) * 0.05
wp = torch.empty(INPROJ_PACKED_ELEMS, dtype=torch.bfloat16, device=dev)
check(lib.mapf_input_proj_pack(_ptr(w), _ptr(wp), None), "pack")
wb = w.to(torch.bfloat16)


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for R in (1, 63, 64, 65, 1000, 12544, 163840):
    x = (torch.randn(R, 784, device=dev) * 0.5).to(torch.bfloat16)
    ref = x.float() @ wb.float().t()
    got = input_proj_rows(x, wp)
    err = (got.float() - ref).abs().max().item()
    lib_out = mm_rows(x, wb)
    err_lib = (lib_out.float() - ref).abs().max().item()
    line = "R=%6d  max |err| %.4f (library %.4f, |ref| max %.2f)" % (R, err, err_lib, ref.abs().max().item())
    if R >= 1000:
        line += "  all rows %.3f ms (library %.3f ms)" % (timed(lambda: input_proj_rows(x, wp, got)), timed(lambda: mm_rows(x, wb)))
        for frac in (0.1, 0.6):
            n = int(R * frac)
            lst = torch.randperm(R, device=dev)[:n].to(torch.int32).contiguous()
            cnt = torch.tensor([n], dtype=torch.int32, device=dev)
            out = torch.zeros((R, 768), dtype=torch.bfloat16, device=dev)
            input_proj_rows(x, wp, out, lst, cnt)
            sel = lst.long()
            ok = torch.equal(out[sel], got[sel]) and int((out != 0).any(dim=1).sum()) <= n
            line += "  | %d%% listed: %s, %.3f ms" % (int(frac * 100), "ok" if ok else "MISMATCH", timed(lambda: input_proj_rows(x, wp, out, lst, cnt)))
    print(line, flush=True)
