#!/usr/bin/env python3
"""A/B harness for the encoder weight-gradient kernel: builds a BASE source (default: the version of
csrc/mapf_wgrad.hip at a git revision, `WGRAD_BASE_REV`, default HEAD~1) and the working-tree csrc/mapf_wgrad.hip as
stand-alone libraries, plus ablated builds of the latter (-DMAPF_WGRAD_ABLATE: 1 no staging, 2 no fragment reads,
3 both = MFMAs only; results wrong, only the time matters), checks the unablated ones against the fp32 weight gradient on
ragged sizes and times everything at the learner's shapes.  `build` runs where hipcc and git are, `run` on the GPU."""
import ctypes
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CAND = os.path.join(ROOT, "mapf_rl_amd", "csrc", "mapf_wgrad.hip")
ABLATE = (1, 2, 3)
NAMES = ["base", "cand"] + ["cand_a%d" % m for m in ABLATE] + [n for n in os.environ.get("WGRAD_EXTRA", "").split(",") if n]  # extras: prebuilt wgrad_ab_<name>.so


def so(name):
    return os.path.join(HERE, "wgrad_ab_%s.so" % name)


def hipcc(src, out, flags=()):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-mllvm", "-pragma-unroll-threshold=262144",
                           "-shared", "-fPIC", "-I" + os.path.join(ROOT, "include")] + list(flags) + [src, "-o", out])


def build():
    rev = os.environ.get("WGRAD_BASE_REV", "HEAD~1")
    base_src = os.path.join(HERE, "wgrad_ab_base.hip")
    with open(base_src, "wb") as f:
        f.write(subprocess.check_output(["git", "-C", ROOT, "show", rev + ":mapf_rl_amd/csrc/mapf_wgrad.hip"]))
    hipcc(base_src, so("base"))
    os.remove(base_src)
    hipcc(CAND, so("cand"))
    for m in ABLATE:
        hipcc(CAND, so("cand_a%d" % m), ["-DMAPF_WGRAD_ABLATE=%d" % m])


def run():
    import torch

    libs = {}
    for name in NAMES:
        lib = ctypes.CDLL(so(name))
        lib.mapf_encoder_wgrad.argtypes = [ctypes.c_void_p] * 2 + [ctypes.c_int64] + [ctypes.c_void_p] * 3  # (round-3 ABI: base revisions from d464af7 on)
        libs[name] = lib.mapf_encoder_wgrad
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for M in (1, 2, 3, 5, 161, 1000, 4321):
        g = torch.Generator(device="cuda").manual_seed(M)
        gz = (torch.randn((M, 7, 7, 128), device="cuda", generator=g) * (torch.rand((M, 7, 7, 128), device="cuda", generator=g) < 0.5)).to(torch.float16)
        a = torch.relu(torch.randn((M, 7, 7, 128), device="cuda", generator=g)).to(torch.float16)
        ref = torch.nn.grad.conv2d_weight(a.float().permute(0, 3, 1, 2), (128, 128, 3, 3), gz.float().permute(0, 3, 1, 2), padding=1)
        for name in [n for n in NAMES if "_a" not in n]:
            ws = torch.full((128, 128, 3, 3, 128), float("nan"), dtype=torch.float32, device="cuda")
            rc = libs[name](gz.data_ptr(), a.data_ptr(), M, None, ws.data_ptr(), st)
            torch.cuda.synchronize()
            got = ws.sum(0).permute(0, 3, 1, 2)
            err = float((got - ref).abs().max()) / max(1.0, float(ref.abs().max()))
            print("M=%6d %s rc=%d finite=%s rel.err=%.2e" % (M, name, rc, bool(torch.isfinite(got).all()), err), flush=True)
    for M in (20736, 138240, 442368):
        gz = (torch.randn((M, 49, 128), device="cuda") * (torch.rand((M, 49, 128), device="cuda") < 0.5)).to(torch.float16)
        a = torch.relu(torch.randn((M, 49, 128), device="cuda")).to(torch.float16)
        ws = torch.empty((128, 128, 9, 128), dtype=torch.float32, device="cuda")
        for name, fn in libs.items():
            for _ in range(2):
                fn(gz.data_ptr(), a.data_ptr(), M, None, ws.data_ptr(), st)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(6):
                fn(gz.data_ptr(), a.data_ptr(), M, None, ws.data_ptr(), st)
            e1.record()
            torch.cuda.synchronize()
            print("M=%6d %-8s %.3f ms per launch" % (M, name, e0.elapsed_time(e1) / 6), flush=True)


if __name__ == "__main__":
    build() if sys.argv[1:2] == ["build"] else run()
