#!/usr/bin/env python3
"""Where does encoder_wgrad_kernel spend its time?  Builds diagnostic variants of csrc/mapf_wgrad.hip with parts
ablated (-DMAPF_WGRAD_ABLATE: 1 no HBM->LDS staging, 2 no LDS fragment reads, 4 no barrier; results are wrong, only
the time matters) and times each on one layer at the learner's shape.  `build` runs where hipcc is, `run` on the GPU."""
import ctypes
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
MODES = [0, 1, 7]


def so(mode):
    return os.path.join(HERE, "wgrad_ablate_%d.so" % mode)


def build():
    for m in MODES:
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-mllvm", "-pragma-unroll-threshold=262144", "-shared", "-fPIC", "-I" + os.path.join(ROOT, "include"),
                               "-DMAPF_WGRAD_ABLATE=%d" % m, os.path.join(ROOT, "mapf_rl_amd", "csrc", "mapf_wgrad.hip"), "-o", so(m)])


def run():
    import torch

    M = 122880
    gz = (torch.randn((M, 49, 128), device="cuda") * (torch.rand((M, 49, 128), device="cuda") < 0.5)).to(torch.bfloat16)
    a = torch.relu(torch.randn((M, 49, 128), device="cuda")).to(torch.bfloat16)
    ws = torch.empty((128, 128, 9, 128), dtype=torch.float32, device="cuda")
    for m in MODES:
        lib = ctypes.CDLL(so(m))
        fn = lib.mapf_encoder_wgrad
        fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        for _ in range(2):
            fn(gz.data_ptr(), a.data_ptr(), M, ws.data_ptr(), st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(6):
            fn(gz.data_ptr(), a.data_ptr(), M, ws.data_ptr(), st)
        e1.record()
        torch.cuda.synchronize()
        print("ablate=%d  %.3f ms per launch" % (m, e0.elapsed_time(e1) / 6), flush=True)


if __name__ == "__main__":
    build() if sys.argv[1:] == ["build"] else run()
