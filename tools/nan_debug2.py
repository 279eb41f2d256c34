import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from mapf_rl_amd.learner import Learner, huber_loss
B, T, N = int(os.environ.get("TB", 192)), 18, 40
torch.manual_seed(0)
lr = Learner(None, device="cuda", batch_size=B)
names = [n for n, _ in lr.model.named_parameters()]
for it in range(8):
    obs = (torch.rand((B, T, N, 6, 9, 9), device="cuda") < 0.3).to(torch.bfloat16)
    hidden = (torch.randn((B * N, 256), device="cuda") * 0.3).half()
    comm = torch.rand((B, T, N, N), device="cuda") < 0.05
    comm |= torch.eye(N, dtype=torch.bool, device="cuda")
    batch = (obs, torch.randint(0, 5, (B, 1), device="cuda"), torch.full((B, 1), -0.075, device="cuda"), torch.zeros((B, 1), device="cuda"),
             torch.full((B, 1), 2.0, device="cuda"), torch.randint(1, 17, (B,), device="cuda"), hidden, comm, None, torch.ones((B, 1), device="cuda"), 0)
    td, q, qn = lr.compute_td(batch)
    loss = (huber_loss(td)).mean()
    lr.bucket.zero()
    loss.backward()
    torch.cuda.synchronize()
    bad = [n for n, p in zip(names, lr.model.parameters()) if not bool(torch.isfinite(p.grad).all())]
    print(it, "loss", float(loss), "td finite", bool(torch.isfinite(td).all()), "nonfinite grads:", bad[:6], "gnorm", float(lr.bucket.flat.norm()), flush=True)
    if bad:
        g = dict(zip(names, [p.grad for p in lr.model.parameters()]))[bad[0]]
        print("   first bad", bad[0], "count nonfinite", int((~torch.isfinite(g)).sum()), "of", g.numel(), flush=True)
