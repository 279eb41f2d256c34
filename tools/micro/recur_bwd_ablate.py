#!/usr/bin/env python3
"""Where does recurrent_bwd_kernel spend its time?  Builds csrc/mapf_recur_bwd.hip with phases ablated (-DMAPF_RBWD_ABLATE:
1 update-cell elementwise, 2 its GEMMs, 4 W_O + attention backward, 8 W_qkv GEMM, 16 recurrent cell; results are wrong,
only the time matters) and times each at the learner's shape.  `build` runs where hipcc is, `run` on the GPU."""
import ctypes
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
MODES = [0, 1, 2, 4, 8, 16, 31]


def so(mode):
    return os.path.join(HERE, "recur_bwd_ablate_%d.so" % mode)


def build():
    for m in MODES:
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-shared", "-fPIC", "-I" + os.path.join(ROOT, "include"),
                               "-DMAPF_RBWD_ABLATE=%d" % m, os.path.join(ROOT, "mapf_rl_amd", "csrc", "mapf_recur_bwd.hip"), "-o", so(m)])


def run():
    import torch

    T, E, N = 16, 192, 40
    R = T * E * N
    bf = torch.bfloat16
    rnd = lambda *s: (torch.rand(s, device="cuda") * 0.5).to(bf)  # noqa: E731
    saves = [rnd(R, 256), rnd(R, 1024), rnd(2, R, 256), rnd(2, R, 384), rnd(2, R, 128), rnd(2, R, 64), rnd(2, R, 1024),
             (torch.rand((2, T * E, 2, 48, 64), device="cuda") / 40).to(bf)]
    comm = (torch.rand((T, E, N, N), device="cuda") < 0.2).to(torch.uint8)
    da0 = rnd(T, E, 256)
    wt = (torch.randn(548864, device="cuda") * 0.03).to(bf)
    outs = [torch.empty((R, 768), dtype=bf, device="cuda"), torch.empty((R, 768), dtype=bf, device="cuda"), torch.empty((2, R, 768), dtype=bf, device="cuda"),
            torch.empty((2, R, 768), dtype=bf, device="cuda"), torch.empty((2, R, 64), dtype=bf, device="cuda"), torch.empty((2, R, 384), dtype=bf, device="cuda"),
            torch.empty((E, 2432), dtype=torch.float32, device="cuda")]
    sp = (ctypes.c_void_p * 8)(*[t.data_ptr() for t in saves])
    op = (ctypes.c_void_p * 7)(*[t.data_ptr() for t in outs])
    vp = ctypes.c_void_p
    for m in MODES:
        fn = ctypes.CDLL(so(m)).mapf_recurrent_backward
        fn.argtypes = [ctypes.POINTER(vp), vp, vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(vp), vp]
        args = (sp, comm.data_ptr(), da0.data_ptr(), wt.data_ptr(), T, E, N, op, None)
        for _ in range(2):
            fn(*args)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            fn(*args)
        e1.record()
        torch.cuda.synchronize()
        print("ablate=%-2d  %.3f ms per launch" % (m, e0.elapsed_time(e1) / 5), flush=True)


if __name__ == "__main__":
    build() if sys.argv[1:] == ["build"] else run()
