#!/usr/bin/env python3
"""Where does recurrent_infer_kernel spend its time?  Diagnostic builds of csrc/mapf_recur.hip with phases removed
(-DMAPF_RECUR_ABLATE: 1 GRU, 2 QKV, 4 attention, 8 W_O, 16 update cell; wrong results, only the time matters), timed on
the actor's shape (4096 environments x 40 agents, one step).  `build` runs where hipcc is, `run` on the GPU."""
import ctypes
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
MODES = [0, 1, 2, 4, 8, 16, 31]


def so(mode):
    return os.path.join(HERE, "recur_ablate_%d.so" % mode)


def build():
    for m in MODES:
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-shared", "-fPIC", "-I" + os.path.join(ROOT, "include"),
                               "-I" + os.path.join(ROOT, "mapf_rl_amd", "csrc"), "-mllvm", "-pragma-unroll-threshold=262144",
                               "-DMAPF_RECUR_ABLATE=%d" % m] + [os.path.join(ROOT, "mapf_rl_amd", "csrc", f) for f in
                               ("mapf_recur.hip", "mapf_recur_nt1.hip", "mapf_recur_nt2.hip", "mapf_recur_wide.hip")] + ["-o", so(m)])


def run():
    import torch

    T, E, N = 1, 4096, int(os.environ.get("ABLATE_AGENTS", "40"))
    gi = (torch.randn((T, E, N, 768), device="cuda") * 0.5).to(torch.bfloat16)
    h0 = (torch.randn((E, N, 256), device="cuda") * 0.3).to(torch.bfloat16)
    comm = (torch.rand((T, E, N, N), device="cuda") < 0.1).to(torch.uint8)
    w = (torch.randn(548864, device="cuda") * 0.05).to(torch.bfloat16)
    b = torch.zeros(3456, device="cuda")
    out = torch.empty((E, N, 256), dtype=torch.bfloat16, device="cuda")
    for m in MODES:
        fn = ctypes.CDLL(so(m)).mapf_recurrent_infer
        fn.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int] * 3 + [ctypes.c_void_p] * 3 + [ctypes.c_int64, ctypes.c_void_p]
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        args = (gi.data_ptr(), h0.data_ptr(), comm.data_ptr(), w.data_ptr(), b.data_ptr(), T, E, N, out.data_ptr(), None, None, 0, st)
        for _ in range(2):
            fn(*args)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            fn(*args)
        e1.record()
        torch.cuda.synchronize()
        print("ablate=%2d  %.3f ms per launch" % (m, e0.elapsed_time(e1) / 5), flush=True)


if __name__ == "__main__":
    build() if sys.argv[1:] == ["build"] else run()
