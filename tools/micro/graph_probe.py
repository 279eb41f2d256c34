#!/usr/bin/env python3
"""HIP-graph probe: does torch.cuda.CUDAGraph capture the library's ctypes-launched kernels next to torch ops, what does a replayed
node cost against an eager launch, and do fork / join side streams inside a capture run concurrently?"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import mapf_rl_amd as M  # noqa: E402


def timed(fn, n=50):
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    host = (time.perf_counter() - t) / n
    torch.cuda.synchronize()
    return host * 1e3, (time.perf_counter() - t) / n * 1e3


def main():
    dev = torch.device("cuda")
    envs = []
    for (E, L, N) in [(512, 10, 1), (512, 15, 2), (512, 20, 3), (512, 25, 4), (512, 30, 5), (512, 35, 6), (512, 40, 6)]:
        env = M.VecEnvironment(E, L, N, device=dev)
        env.reset_envs(None, 0.3, seed=1)
        envs.append((env, torch.zeros((E, N), dtype=torch.int8, device=dev)))
    K = 10

    def body():
        for _ in range(K):
            for env, a in envs:
                env.step(a)

    body()
    print("eager   : host %.3f ms, wall %.3f ms per %d launches" % (*timed(body), K * len(envs)))
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        body()
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        body()
    print("graph   : host %.3f ms, wall %.3f ms per %d nodes (serial chain)" % (*timed(g.replay), K * len(envs)))

    # fork / join: every level on its own stream inside the capture
    sides = [torch.cuda.Stream() for _ in envs]
    g2 = torch.cuda.CUDAGraph()

    def forked():
        cur = torch.cuda.current_stream()
        for (env, a), sd in zip(envs, sides):
            sd.wait_stream(cur)
            with torch.cuda.stream(sd):
                for _ in range(K):
                    env.step(a)
        for sd in sides:
            cur.wait_stream(sd)

    forked()
    torch.cuda.synchronize()
    print("eager forked: host %.3f ms, wall %.3f ms" % timed(forked))
    with torch.cuda.graph(g2):
        forked()
    print("graph forked: host %.3f ms, wall %.3f ms per %d nodes in %d branches" % (*timed(g2.replay), K * len(envs), len(envs)))
    for env, _ in envs:
        env.check_status()

    # torch ops + allocation inside the capture
    x = torch.randn(4096, 784, device=dev, dtype=torch.bfloat16)
    w = torch.randn(768, 784, device=dev, dtype=torch.bfloat16)
    g3 = torch.cuda.CUDAGraph()

    def mixed():
        y = torch.mm(x, w.t())
        z = torch.relu(y).float().sum(dim=1)
        envs[0][0].step(envs[0][1])
        return z

    mixed()
    torch.cuda.synchronize()
    with torch.cuda.graph(g3):
        z = mixed()
    g3.replay()
    torch.cuda.synchronize()
    print("mixed graph ok, z[0] = %.3f (eager %.3f)" % (float(z[0]), float(mixed()[0])))
    print("eager mixed: host %.3f wall %.3f;  graph mixed: host %.3f wall %.3f" % (*timed(mixed), *timed(g3.replay)))


if __name__ == "__main__":
    main()
