import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import config
from mapf_rl_amd.curriculum import CurriculumActors
from mapf_rl_amd.learner import Learner
from mapf_rl_amd.replay import GlobalBuffer
CurriculumActors.GRAPH = False
torch.manual_seed(0)
buf = GlobalBuffer(16384, max_agents=6, init_set=(1, 10), max_map_length=40, pass_rate=0.9)
lr = Learner(buf, device="cuda", batch_size=192)
cur = CurriculumActors(lr.model, buf, envs_per_level=512, seed=0, reward_fn=config.reward_fn, weights_period=400)
for i in range(300):
    cur.step()
torch.cuda.synchronize()
bad = 0
for trial in range(300):
    batch = buf.sample_batch(192)
    pl = lr._fused.plan(batch)
    pl["event"].synchronize()
    h = pl["host"].numpy()
    po, pt = pl["online"], pl["target"]
    dup = po.dup.cpu().long()
    for k, p in enumerate((po, pt)):
        nact, slot = p.nact.cpu(), p.slot.cpu().long()
        need = (slot.unsqueeze(0) >= 0) & (slot.unsqueeze(0) < nact.unsqueeze(-1))
        ar = torch.arange(p.T).view(-1, 1, 1)
        flagged = (need & (dup[:p.T] >= ar)).sum(dim=(0, 2))
        want = torch.from_numpy(h[4 + k]).long()
        if not torch.equal(flagged, want):
            bad += 1
            b = int((flagged != want).nonzero()[0])
            print("trial", trial, "set", k, "window", b, "flagged", int(flagged[b]), "ucnt", int(want[b]), "nact", nact[:, b].tolist(), "slot", slot[b].tolist(),
                  "dup col", dup[:, b, :].t().tolist()[:2], flush=True)
            if bad > 5:
                sys.exit(0)
print("mismatching (batch, set) pairs:", bad)
