"""One rank of the multi-rank update tests (tests/test_learner_gpu.py): the eager launch sequence and the graph-replayed one, both
with the gradient exchange inside, on this rank's own replay; prints one JSON line of error measures and a parameter checksum.

    RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT from the environment; argv: backend (gloo | nccl), number of updates.
All ranks use GPU 0 (the test box has one).  With WORLD_SIZE = 1 the one-rank group is treated as several ranks
(learner.FORCE_EXCHANGE): the collective calls run -- RCCL when the backend is nccl -- and average over one rank."""
import json
import os
import sys

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    backend, n = sys.argv[1], int(sys.argv[2])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=dev)
    else:
        dist.init_process_group(backend)
    from mapf_rl_amd import learner as learner_mod
    from mapf_rl_amd.learner import Learner
    from mapf_rl_amd.model import Network
    from mapf_rl_amd.update import FusedUpdate
    from tests.test_learner_gpu import _filled_replay

    if world == 1:
        learner_mod.FORCE_EXCHANGE = True
    calls = {"begin": 0, "finish": 0}

    def run(graph, n_updates):
        saved, FusedUpdate.GRAPH = FusedUpdate.GRAPH, graph
        try:
            buf = _filled_replay(seed=1 + rank)       # every rank its own episodes
            torch.manual_seed(7)                      # ... and the same initial weights
            torch.cuda.manual_seed(7 + rank)
            lr = Learner(buf, device="cuda", batch_size=48, model=Network())
            cls = type(lr.bucket)
            if not hasattr(cls, "_counted"):
                b0, f0 = cls.begin, cls.finish

                def begin(self, lo, hi, group=None):
                    calls["begin"] += 1
                    return b0(self, lo, hi, group)

                def finish(self, group=None):
                    calls["finish"] += 1
                    return f0(self, group)

                cls.begin, cls.finish, cls._counted = begin, finish, True
            outs = []
            for _ in range(n_updates):
                o = lr.update()
                outs.append({k: v.detach().clone() for k, v in o.items()})
            torch.cuda.synchronize()
            fu = lr._fused
            return dict(outs=outs, params=[p.detach().clone() for p in lr.model.parameters()], replays=fu.graph_replays, captures=fu.graph_captures,
                        step=fu.flat.step, grads={k: fu.flat.mem(fu.flat.grads, k).clone() for k in fu.flat.names}, graph_mode=fu.graph_mode())
        finally:
            FusedUpdate.GRAPH = saved

    p0 = run(False, 0)["params"]
    e1, g1 = run(False, 1), run(True, 1)
    grad_err = max(float((e1["grads"][k] - g1["grads"][k]).norm()) / (float(e1["grads"][k].norm()) + 1e-7) for k in e1["grads"])
    calls["begin"] = calls["finish"] = 0
    e, g = run(False, n), run(True, n)
    a, b = e["outs"][0], g["outs"][0]
    num = sum(float((x - y).pow(2).sum()) for x, y in zip(e["params"], g["params"]))
    den = sum(float((x - y).pow(2).sum()) for x, y in zip(e["params"], p0))
    # every rank must hold the same parameters after the same number of exchanged updates: compare a checksum across ranks
    flat = torch.cat([p.reshape(-1).double() for p in g["params"]])
    w = torch.arange(1, flat.numel() + 1, device=flat.device, dtype=torch.float64)
    sums = torch.stack([flat.sum(), (flat * w).sum() / flat.numel(), flat.abs().sum()])
    gathered = [torch.zeros_like(sums) for _ in range(world)]
    dist.all_gather(gathered, sums)
    same = all(bool(torch.equal(gathered[0], t)) for t in gathered)
    out = dict(rank=rank, world=world, backend=dist.get_backend(), graph_mode_on=bool(g["graph_mode"]), replays_eager=e["replays"], replays_graph=g["replays"],
               captures=g["captures"], steps=(e["step"], g["step"]), grad_err=grad_err,
               td_err=float((a["td"] - b["td"]).abs().max()), q_err=float((a["q"] - b["q"]).abs().max()),
               loss=(float(a["loss"]), float(b["loss"])), grad_norm=(float(a["grad_norm"]), float(b["grad_norm"])),
               param_diff_over_travel=num / max(den, 1e-30), same_params_on_all_ranks=same, exchange_calls=calls,
               finite=all(bool(torch.isfinite(v).all()) for o in g["outs"] for v in o.values()))
    print(json.dumps(out), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
