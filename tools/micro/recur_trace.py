#!/usr/bin/env python3
"""Cycle stamps inside recurrent_infer_kernel: a diagnostic build of csrc/mapf_recur.hip (-DMAPF_RECUR_TRACE) in which wave 0 of one
workgroup records s_memtime at every phase barrier (before / after) and at five points inside each GRU cell; prints the
durations between consecutive stamps for one step.  `build` runs where hipcc is, `run [agents] [envs]` on the GPU."""
import ctypes
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "mapf_rl_amd", "csrc")
SO = os.path.join(HERE, "recur_trace.so")
NAMES = {30: "kernel entry", 31: "prologue: LDS zero fill, biases, h0 | barrier", 32: "h_out stored (issued)", 1: "gi requests, mask+ridx | barrier", 2: "GRU cell | barrier", 3: "QKV | barrier", 6: "attention (scores, softmax, ctx in registers) | barrier",
         7: "W_O | barrier", 8: "update cell | barrier", 20: "cell: entered", 21: "cell: block A MFMAs + reloads issued", 22: "cell: block A pointwise",
         23: "cell: block B MFMAs", 24: "cell: block B pointwise"}


def build():
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-shared", "-fPIC", "-mllvm", "-pragma-unroll-threshold=262144",
                           "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-DMAPF_RECUR_TRACE", "-DMAPF_RECUR_TRACE_WG=%s" % os.environ.get("TRACE_WG", "300")] +
                          [os.path.join(CSRC, f) for f in ("mapf_recur.hip", "mapf_recur_nt1.hip", "mapf_recur_nt2.hip", "mapf_recur_wide.hip")] + ["-o", SO])


def run():
    import torch

    N = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    E = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
    T = 1
    lib = ctypes.CDLL(SO)
    fn = lib.mapf_recurrent_infer
    fn.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int] * 3 + [ctypes.c_void_p] * 3 + [ctypes.c_int64, ctypes.c_void_p]
    reader = getattr(lib, "mapf_recur_trace_read_nt%d" % (1 if N <= 16 else 2 if N <= 32 else 3))
    g = torch.Generator(device="cuda").manual_seed(0)
    gi = (torch.randn((T, E, N, 768), device="cuda", generator=g) * 0.5).to(torch.bfloat16)
    h0 = (torch.randn((E, N, 256), device="cuda", generator=g) * 0.3).to(torch.bfloat16)
    comm = (torch.rand((T, E, N, N), device="cuda", generator=g) < 0.1).to(torch.uint8)
    w = (torch.randn(548864, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
    b = torch.zeros(3456, device="cuda")
    out = torch.empty((E, N, 256), dtype=torch.bfloat16, device="cuda")
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    buf = (ctypes.c_ulonglong * 128)()
    rowidx, nrows = None, 0
    if os.environ.get("NOGI"):  # no agent has an input projection row: the first cell without its gi loads
        rowidx, nrows = torch.full((T, E, N), -1, dtype=torch.int32, device="cuda"), 1
    elif os.environ.get("SAMEGI"):  # every agent reads row 0
        rowidx, nrows = torch.zeros((T, E, N), dtype=torch.int32, device="cuda"), 1
    for it in range(3):
        torch.cuda.synchronize()
        reader(buf, 1)
        fn(gi.data_ptr(), h0.data_ptr(), comm.data_ptr(), w.data_ptr(), b.data_ptr(), T, E, N, out.data_ptr(), None, rowidx.data_ptr() if rowidx is not None else None, nrows, st)
        torch.cuda.synchronize()
    n = reader(buf, 1)
    stamps = [(buf[i] >> 56, buf[i] & ((1 << 56) - 1)) for i in range(n)]
    print("%d agents, %d environments: %d stamps, workgroup = %d cycles" % (N, E, n, stamps[-1][1] - stamps[0][1]))
    for (i0, c0), (i1, c1) in zip(stamps, stamps[1:]):
        what = NAMES.get(i1, "barrier wait (%s)" % NAMES.get(i1 - 100, "?").split(" |")[0]) if i1 < 100 else "   wait at the barrier behind: " + NAMES[i1 - 100].split(" |")[0]
        print("%8d cycles  -> %s" % (c1 - c0, what))


if __name__ == "__main__":
    build() if sys.argv[1:2] == ["build"] else run()
