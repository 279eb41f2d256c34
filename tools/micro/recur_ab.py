#!/usr/bin/env python3
"""A/B harness for the narrow recurrence kernel: builds csrc/mapf_recur.hip at a git revision (`RECUR_BASE_REV`, default HEAD)
and from the working tree as stand-alone libraries, compares mapf_recurrent_infer's outputs on the same inputs and times it at
the actor's shape (1 step x 4096 environments x 40 agents) and the learner's (18 steps x 192 x 40 and x 6).
`build` runs where hipcc and git are, `run` on the GPU."""
import ctypes
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SRC = os.path.join(ROOT, "mapf_rl_amd", "csrc", "mapf_recur.hip")


def so(name):
    return os.path.join(HERE, "recur_ab_%s.so" % name)


CSRC = os.path.join(ROOT, "mapf_rl_amd", "csrc")
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-mllvm", "-pragma-unroll-threshold=262144", "-shared", "-fPIC",
         "-I" + os.path.join(ROOT, "include")]


def hipcc(srcdir, out):
    """mapf_recur.hip + its one- and two-tile builds from `srcdir`, the wide kernel (the N > 48 entry they forward to) from the tree"""
    extra = os.environ.get("RECUR_CAND_FLAGS", "").split() if out.endswith("cand.so") else []
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + extra + ["-I" + srcdir, "-I" + CSRC] +
                          [os.path.join(srcdir, f) for f in ("mapf_recur.hip", "mapf_recur_nt1.hip", "mapf_recur_nt2.hip")] +
                          [os.path.join(CSRC, "mapf_recur_wide.hip"), "-o", out])


def build():
    import shutil
    import tempfile

    rev = os.environ.get("RECUR_BASE_REV", "HEAD")
    tmp = tempfile.mkdtemp(prefix="recur_ab_")
    for f in ("mapf_recur.hip", "mapf_recur_nt1.hip", "mapf_recur_nt2.hip", "mapf_recur_internal.h"):
        with open(os.path.join(tmp, f), "wb") as fh:
            fh.write(subprocess.check_output(["git", "-C", ROOT, "show", rev + ":mapf_rl_amd/csrc/" + f]))
    hipcc(tmp, so("base"))
    shutil.rmtree(tmp)
    hipcc(CSRC, so("cand"))


def run():
    import torch

    fns = {}
    for name in ("base", "cand"):
        fn = ctypes.CDLL(so(name)).mapf_recurrent_infer
        fn.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int] * 3 + [ctypes.c_void_p] * 3 + [ctypes.c_int64, ctypes.c_void_p]
        fns[name] = fn
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    g = torch.Generator(device="cuda").manual_seed(0)
    w = (torch.randn(548864, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
    b = torch.randn(3456, device="cuda", generator=g) * 0.1
    for T, E, N in ((1, 4096, 40), (18, 192, 40), (18, 192, 6), (1, 4096, 6), (1, 1400, 16), (3, 33, 17), (18, 192, 24)):
        gi = (torch.randn((T, E, N, 768), device="cuda", generator=g) * 0.5).to(torch.bfloat16)
        h0 = (torch.randn((E, N, 256), device="cuda", generator=g) * 0.3).to(torch.bfloat16)
        comm = (torch.rand((T, E, N, N), device="cuda", generator=g) < 0.1).to(torch.uint8)
        comm |= torch.eye(N, device="cuda", dtype=torch.uint8)
        outs = {}
        for name, fn in fns.items():
            out = torch.zeros((E, N, 256), dtype=torch.bfloat16, device="cuda")
            a0 = torch.zeros((T, E, 256), dtype=torch.bfloat16, device="cuda")
            args = (gi.data_ptr(), h0.data_ptr(), comm.data_ptr(), w.data_ptr(), b.data_ptr(), T, E, N, out.data_ptr(), a0.data_ptr(), None, 0, st)
            for _ in range(2):
                rc = fn(*args)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                fn(*args)
            e1.record()
            torch.cuda.synchronize()
            outs[name] = (out, a0)
            print("T=%2d E=%4d N=%2d %-5s rc=%d %.3f ms per launch" % (T, E, N, name, rc, e0.elapsed_time(e1) / 5), flush=True)
        if T > 1:  # the training forward (saves what the backward needs)
            R = T * E * N
            sizes = [R * 256, R * 1024, 2 * R * 256, 2 * R * 384, 2 * R * 128, 2 * R * 64, 2 * R * 1024, 2 * T * E * 2 * 48 * 64]
            for name in ("base", "cand"):
                fs = ctypes.CDLL(so(name)).mapf_recurrent_forward_save
                fs.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int] * 3 + [ctypes.c_void_p] * 4 + [ctypes.c_int64, ctypes.c_void_p]
                bufs = [torch.zeros(n, dtype=torch.bfloat16, device="cuda") for n in sizes]
                ptrs = (ctypes.c_void_p * 8)(*[b_.data_ptr() for b_ in bufs])
                out = torch.zeros((E, N, 256), dtype=torch.bfloat16, device="cuda")
                a0 = torch.zeros((T, E, 256), dtype=torch.bfloat16, device="cuda")
                args = (gi.data_ptr(), h0.data_ptr(), comm.data_ptr(), w.data_ptr(), b.data_ptr(), T, E, N, out.data_ptr(), a0.data_ptr(), ptrs, None, 0, st)
                for _ in range(2):
                    rc = fs(*args)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    fs(*args)
                e1.record()
                torch.cuda.synchronize()
                print("T=%2d E=%4d N=%2d %-5s rc=%d %.3f ms per forward_save launch" % (T, E, N, name, rc, e0.elapsed_time(e1) / 5), flush=True)
        d = (outs["base"][0].float() - outs["cand"][0].float()).abs()
        print("   max |diff| %.4g, differing elements %.4g, nan base/cand %d/%d" % (d.max().item(), (d > 0).float().mean().item(), outs["base"][0].isnan().sum().item(), outs["cand"][0].isnan().sum().item()))
        print("   hidden bit-identical=%s  agent-0 trace bit-identical=%s" % (torch.equal(outs["base"][0], outs["cand"][0]), torch.equal(outs["base"][1], outs["cand"][1])), flush=True)


if __name__ == "__main__":
    build() if sys.argv[1:2] == ["build"] else run()
