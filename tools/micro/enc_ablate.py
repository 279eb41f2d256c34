#!/usr/bin/env python3
"""What does saving cost the training forward?  (Also: 4 = conflict-free LDS rows, 8 = conv0 on zeros -- NOT a measure of its input
gather: all-zero activations let the chip clock higher, the whole kernel runs 23 % faster --, 64 = one gather per k-step: the gather is 2 %.)
  Builds csrc/mapf_encoder.hip with -DMAPF_ENC_ABLATE (1 = no saved-activation
copies, 2 = no ReLU sign words; outputs are incomplete, only the time matters) and times mapf_encoder_forward_save next to
mapf_encoder_forward at the learner's shape.  `build` runs where hipcc is, `run` on the GPU."""
import ctypes
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
# round 6: MODES from the environment (default: the round-4 set); mode 1000 = no ablation, saved activations through NON-TEMPORAL stores
MODES = [int(m) for m in os.environ.get("MODES", "0,1,2,3,4,8,64").split(",")]


def so(mode):
    return os.path.join(HERE, "enc_ablate_%d.so" % mode)


def build():
    for m in MODES:
        flags = ["-DMAPF_ENC_ABLATE=0", "-DMAPF_ENC_NT_SAVE=1"] if m == 1000 else ["-DMAPF_ENC_ABLATE=%d" % m]
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-mllvm", "-pragma-unroll-threshold=262144", "-shared", "-fPIC", "-I" + os.path.join(ROOT, "include")] +
                              flags + [os.path.join(ROOT, "mapf_rl_amd", "csrc", "mapf_encoder.hip"), "-o", so(m)])


def run():
    import torch

    M = int(os.environ.get("ROWS", "122880"))
    obs = (torch.rand((M, 6, 9, 9), device="cuda") < 0.3).to(torch.uint8)
    w = (torch.randn(894976, device="cuda") * 0.03).to(torch.bfloat16)
    b = torch.zeros(912, dtype=torch.float32, device="cuda")
    lat = torch.empty((M, 784), dtype=torch.bfloat16, device="cuda")
    acts = torch.empty((7, M, 49, 128), dtype=torch.bfloat16, device="cuda")
    bits = torch.empty((7, M, 49, 4), dtype=torch.int32, device="cuda")
    vp = ctypes.c_void_p
    for m in MODES * int(os.environ.get("TURNS", "1")):  # (TURNS > 1: the modes in turns -- the first library of a process runs on a cold chip)
        lib = ctypes.CDLL(so(m))
        for name in ("mapf_encoder_forward", "mapf_encoder_forward_save"):
            fn = getattr(lib, name)
            save = name.endswith("save")
            fn.argtypes = [vp, ctypes.c_int, ctypes.c_int64, vp, vp, vp] + ([vp, vp] if save else []) + [vp]
            args = [obs.data_ptr(), 0, M, w.data_ptr(), b.data_ptr(), lat.data_ptr()] + ([acts.data_ptr(), bits.data_ptr()] if save else []) + [None]
            for _ in range(2):
                fn(*args)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                fn(*args)
            e1.record()
            torch.cuda.synchronize()
            print("rows=%d ablate=%d  %-26s %.3f ms  (%.1f ns per observation)" % (M, m, name, e0.elapsed_time(e1) / 5, e0.elapsed_time(e1) / 5 * 1e6 / M), flush=True)


if __name__ == "__main__":
    build() if sys.argv[1:] == ["build"] else run()
