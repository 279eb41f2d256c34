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


def hipcc(src, out):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3"] + os.environ.get("RECUR_CAND_FLAGS", "").split() * (out.endswith("cand.so")) + [ "--offload-arch=gfx950", "-std=c++17", "-mllvm", "-pragma-unroll-threshold=262144",
                           "-shared", "-fPIC", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "mapf_rl_amd", "csrc"), src,
                           os.path.join(ROOT, "mapf_rl_amd", "csrc", "mapf_recur_wide.hip"), "-o", out])  # (the wide kernel: the N > 48 entry it forwards to)


def build():
    rev = os.environ.get("RECUR_BASE_REV", "HEAD")
    base_src = os.path.join(HERE, "recur_ab_base.hip")
    with open(base_src, "wb") as f:
        f.write(subprocess.check_output(["git", "-C", ROOT, "show", rev + ":mapf_rl_amd/csrc/mapf_recur.hip"]))
    hipcc(base_src, so("base"))
    os.remove(base_src)
    hipcc(SRC, so("cand"))


def run():
    import torch

    fns = {}
    for name in ("base", "cand"):
        fn = ctypes.CDLL(so(name)).mapf_recurrent_infer
        fn.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int] * 3 + [ctypes.c_void_p] * 3
        fns[name] = fn
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    g = torch.Generator(device="cuda").manual_seed(0)
    w = (torch.randn(548864, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
    b = torch.randn(3456, device="cuda", generator=g) * 0.1
    for T, E, N in ((1, 4096, 40), (18, 192, 40), (18, 192, 6), (3, 33, 17)):
        gi = (torch.randn((T, E, N, 768), device="cuda", generator=g) * 0.5).to(torch.bfloat16)
        h0 = (torch.randn((E, N, 256), device="cuda", generator=g) * 0.3).to(torch.bfloat16)
        comm = (torch.rand((T, E, N, N), device="cuda", generator=g) < 0.1).to(torch.uint8)
        comm |= torch.eye(N, device="cuda", dtype=torch.uint8)
        outs = {}
        for name, fn in fns.items():
            out = torch.zeros((E, N, 256), dtype=torch.bfloat16, device="cuda")
            a0 = torch.zeros((T, E, 256), dtype=torch.bfloat16, device="cuda")
            args = (gi.data_ptr(), h0.data_ptr(), comm.data_ptr(), w.data_ptr(), b.data_ptr(), T, E, N, out.data_ptr(), a0.data_ptr(), st)
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
        print("   hidden bit-identical=%s  agent-0 trace bit-identical=%s" % (torch.equal(outs["base"][0], outs["cand"][0]), torch.equal(outs["base"][1], outs["cand"][1])), flush=True)


if __name__ == "__main__":
    build() if sys.argv[1:2] == ["build"] else run()
