import torch, time, torch.nn as nn, torch.nn.functional as F
torch.manual_seed(0)
dev = "cuda"
def bench(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.time() - t) / n
for M in (32768,):
    for fmt in ("nchw", "nhwc"):
        conv = nn.Conv2d(128, 128, 3, 1, 1).to(dev).to(torch.bfloat16)
        x = torch.randn(M, 128, 7, 7, device=dev, dtype=torch.bfloat16)
        if fmt == "nhwc":
            conv = conv.to(memory_format=torch.channels_last); x = x.contiguous(memory_format=torch.channels_last)
        flops = 2 * M * 49 * 128 * 1152
        with torch.no_grad():
            t = bench(lambda: conv(x))
        print(fmt, "M", M, "fwd %.2f ms  %.0f TFLOP/s" % (t * 1e3, flops / t / 1e12), flush=True)
        xg = x.clone().requires_grad_(True)
        def fb():
            y = conv(xg); y.backward(torch.ones_like(y)); 
        t2 = bench(fb, 5)
        print(fmt, "fwd+bwd %.2f ms  %.0f TFLOP/s (3x flops)" % (t2 * 1e3, 3 * flops / t2 / 1e12), flush=True)
    # plain GEMM ceiling at the implicit-GEMM shape
    a = torch.randn(M * 49, 1152, device=dev, dtype=torch.bfloat16); b = torch.randn(1152, 128, device=dev, dtype=torch.bfloat16)
    t = bench(lambda: a @ b)
    print("gemm [%d x 1152] @ [1152 x 128]: %.2f ms %.0f TFLOP/s" % (M * 49, t * 1e3, 2 * M * 49 * 1152 * 128 / t / 1e12), flush=True)
