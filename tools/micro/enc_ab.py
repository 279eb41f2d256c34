#!/usr/bin/env python3
"""A/B harness for the fused encoder kernels: builds csrc/mapf_encoder.hip at a git revision (`ENC_BASE_REV`, default HEAD)
and from the working tree as stand-alone libraries, compares their outputs on the same inputs (latents, saved layers,
ReLU sign words, backward gradients) and times forward / forward_save / backward at the learner's and the actor's
shapes.  `build` runs where hipcc and git are, `run` on the GPU."""
import ctypes
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CAND = os.path.join(ROOT, "mapf_rl_amd", "csrc", "mapf_encoder.hip")
vp = ctypes.c_void_p


def so(name):
    return os.path.join(HERE, "enc_ab_%s.so" % name)


def hipcc(src, out):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-mllvm", "-pragma-unroll-threshold=262144",
                           "-shared", "-fPIC", "-I" + os.path.join(ROOT, "include"), src, "-o", out])


def build():
    rev = os.environ.get("ENC_BASE_REV", "HEAD")
    base_src = os.path.join(HERE, "enc_ab_base.hip")
    with open(base_src, "wb") as f:
        f.write(subprocess.check_output(["git", "-C", ROOT, "show", rev + ":mapf_rl_amd/csrc/mapf_encoder.hip"]))
    hipcc(base_src, so("base"))
    os.remove(base_src)
    hipcc(CAND, so("cand"))


def run():
    import torch

    libs = {n: ctypes.CDLL(so(n)) for n in ("base", "cand")}
    for lib in libs.values():
        lib.mapf_encoder_forward.argtypes = [vp, ctypes.c_int, ctypes.c_int64, vp, vp, vp, vp]
        lib.mapf_encoder_forward_save.argtypes = [vp, ctypes.c_int, ctypes.c_int64, vp, vp, vp, vp, vp, vp]
        lib.mapf_encoder_backward.argtypes = [vp, vp, ctypes.c_int64, vp, vp, vp, vp, vp, vp, vp, vp]  # (round-3 ABI: base revisions from d464af7 on)
    g = torch.Generator(device="cuda").manual_seed(1)
    w = (torch.randn(894976, device="cuda", generator=g) * 0.03).to(torch.float16)
    wt = (torch.randn(888832, device="cuda", generator=g) * 0.03).to(torch.float16)
    b = torch.randn(912, dtype=torch.float32, device="cuda", generator=g) * 0.1

    def timed(fn, n=5):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n

    for M in (1001, 138240, 163840):
        obs = (torch.rand((M, 6, 9, 9), device="cuda", generator=g) < 0.3).to(torch.uint8)
        gl = (torch.randn((M, 784), device="cuda", generator=g)).to(torch.bfloat16)
        out = {}
        for name, lib in libs.items():
            lat = torch.zeros((M, 784), dtype=torch.bfloat16, device="cuda")
            lat2 = torch.zeros((M, 784), dtype=torch.bfloat16, device="cuda")
            acts = torch.zeros((7, M, 49, 128), dtype=torch.float16, device="cuda")
            bits = torch.zeros((7, M, 49, 4), dtype=torch.int32, device="cuda")
            gz = torch.zeros((7, M, 49, 128), dtype=torch.float16, device="cuda")
            gz7 = torch.zeros((M, 49, 16), dtype=torch.float16, device="cuda")
            sw = torch.zeros(2, dtype=torch.int32, device="cuda")
            nblk = (M + 3) // 4
            gbp = torch.zeros((7, nblk, 128), dtype=torch.float32, device="cuda")
            gb7 = torch.zeros((4 * nblk, 16), dtype=torch.float32, device="cuda")
            f = lambda: lib.mapf_encoder_forward(obs.data_ptr(), 0, M, w.data_ptr(), b.data_ptr(), lat.data_ptr(), None)
            fs = lambda: lib.mapf_encoder_forward_save(obs.data_ptr(), 0, M, w.data_ptr(), b.data_ptr(), lat2.data_ptr(), acts.data_ptr(), bits.data_ptr(), None)
            bw = lambda: lib.mapf_encoder_backward(gl.data_ptr(), lat2.data_ptr(), M, bits.data_ptr(), wt.data_ptr(), gz.data_ptr(), gbp.data_ptr(),
                                                   gz7.data_ptr(), gb7.data_ptr(), sw.data_ptr(), None)
            tf, tfs, tb = timed(f), timed(fs), timed(bw)
            print("M=%6d %-5s forward %.3f ms   forward_save %.3f ms   backward %.3f ms" % (M, name, tf, tfs, tb), flush=True)
            out[name] = (lat, lat2, acts, bits, gz, gz7)
        for k, nm in enumerate(("latent", "latent(save)", "saved layers", "relu bits", "gz", "gz7")):
            a, c = out["base"][k], out["cand"][k]
            same = torch.equal(a, c)
            d = 0.0 if same else float((a.float() - c.float()).abs().max())
            print("   %-14s bit-identical=%s max|diff|=%.3g" % (nm, same, d), flush=True)


if __name__ == "__main__":
    build() if sys.argv[1:2] == ["build"] else run()
