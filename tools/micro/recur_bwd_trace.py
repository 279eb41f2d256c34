#!/usr/bin/env python3
"""Cycle stamps inside recurrent_bwd_kernel (see tools/micro/recur_trace.py): wave 0 of one workgroup stamps every phase barrier of the
backward-through-time kernel, before and after.  Random saved tensors (the time does not depend on the values).
`build` runs where hipcc is, `run [agents] [envs] [steps]` on the GPU."""
import ctypes
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "mapf_rl_amd", "csrc")
SO = os.path.join(HERE, os.environ.get("BT_SO", "recur_bwd_trace.so"))
PHASES = ["", "external gradient, partner counts", "mask counts", "update flags", "(1) update-cell elementwise", "(2) U_hh^T, U_ih^T GEMMs", "(3) W_O^T",
          "(4) head 0: images", "head 0: dP", "head 0: softmax backward", "head 0: dv dq dk", "(5) W_qkv^T", "recurrent cell elementwise", "W_hh^T GEMM"]


def build():
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-shared", "-fPIC", "-mllvm", "-pragma-unroll-threshold=262144",
                           "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-DMAPF_RECUR_TRACE", "-DMAPF_RECUR_TRACE_WG=%s" % os.environ.get("TRACE_WG", "100")] +
                          [os.path.join(CSRC, f) for f in ("mapf_recur_bwd.hip", "mapf_recur_bwd_nt1.hip", "mapf_recur_bwd_nt2.hip", "mapf_recur_wide_bwd.hip")] + ["-o", SO])


def run():
    import torch

    N = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    E = int(sys.argv[3]) if len(sys.argv) > 3 else 192
    T = int(sys.argv[4]) if len(sys.argv) > 4 else 18
    lib = ctypes.CDLL(SO)
    fn = lib.mapf_recurrent_backward
    fn.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 3 + [ctypes.c_void_p] * 2 + [ctypes.c_int64, ctypes.c_void_p]
    reader = getattr(lib, "mapf_recur_btrace_read_nt%d" % (1 if N <= 16 else 2 if N <= 32 else 3))
    g = torch.Generator(device="cuda").manual_seed(0)
    R = T * E * N
    rnd = lambda n, s=0.3: (torch.rand(n, device="cuda", generator=g) * s).to(torch.bfloat16)
    saved = [rnd(R * 256), rnd(R * 1024, 0.9), rnd(2 * R * 256), rnd(2 * R * 384), rnd(2 * R * 128), rnd(2 * R * 64), rnd(2 * R * 1024, 0.9), rnd(2 * T * E * 2 * 48 * 64, 0.05)]
    comm = (torch.rand((T, E, N, N), device="cuda", generator=g) < 0.1).to(torch.uint8)
    comm |= torch.eye(N, device="cuda", dtype=torch.uint8)
    dA0 = rnd(T * E * 256, 0.01)
    wt = (torch.randn(548864, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
    outs = [torch.zeros(R * 768, dtype=torch.bfloat16, device="cuda") for _ in range(2)] + [torch.zeros(2 * R * 768, dtype=torch.bfloat16, device="cuda") for _ in range(2)] + \
           [torch.zeros(2 * R * 64, dtype=torch.bfloat16, device="cuda"), torch.zeros(2 * R * 384, dtype=torch.bfloat16, device="cuda"), torch.zeros(E * 2432, dtype=torch.float32, device="cuda")]
    sp = (ctypes.c_void_p * 8)(*[x.data_ptr() for x in saved])
    op = (ctypes.c_void_p * 7)(*[x.data_ptr() for x in outs])
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    buf = (ctypes.c_ulonglong * 128)()
    for it in range(3):
        torch.cuda.synchronize()
        reader(buf, 1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = fn(sp, comm.data_ptr(), dA0.data_ptr(), wt.data_ptr(), T, E, N, op, None, 0, st)
        e1.record()
        torch.cuda.synchronize()
    n = reader(buf, 1)
    stamps = [(buf[i] >> 56, buf[i] & ((1 << 56) - 1)) for i in range(n)]
    print("rc=%d, %d agents, %d environments, %d steps: launch %.3f ms; %d stamps (the first steps walked)" % (rc, N, E, T, e0.elapsed_time(e1), n))
    # barrier k of a step: 1-3 prologue, then per round 4, 5, 6, then per head 7..10, then 11; after both rounds 12, 13
    names = {1: PHASES[1], 2: PHASES[2], 3: PHASES[3], 4: PHASES[4], 5: PHASES[5], 6: PHASES[6], 7: "(4) images of a head", 8: "dP", 9: "softmax backward", 10: "dv dq dk",
             11: PHASES[11], 12: PHASES[12], 13: PHASES[13]}
    for (i0, c0), (i1, c1) in zip(stamps, stamps[1:]):
        if i1 < 100:
            print("%8d cycles  %s" % (c1 - c0, names.get(i1, "?")))
        else:
            print("%8d cycles     barrier wait" % (c1 - c0))


if __name__ == "__main__":
    build() if sys.argv[1:2] == ["build"] else run()
