import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import mapf_rl_amd as M
from mapf_rl_amd.actor import VecActor
from mapf_rl_amd.learner import Learner, huber_loss
from mapf_rl_amd.replay import GlobalBuffer
E, L, N, B = int(os.environ.get('TE',128)), 32, 40, int(os.environ.get('TB',16))
if os.environ.get('TBENCH'): torch.backends.cudnn.benchmark = True
import time
maps, agents, goals, _ = M.generate_scenarios(E, L, N, 0.3, seed=1)
env = M.VecEnvironment(E, L, N); env.load(maps, agents, goals)
buf = GlobalBuffer(int(os.environ.get('TCAP',256)), max_agents=N)
lr = Learner(buf, device="cuda", batch_size=B)
actor = VecActor(env, lr.model, buf, max_steps=24, seed=0, density=0.3, keep_flushed=True)
for i in range(30):
    actor.step()
torch.cuda.synchronize()
ep = actor.flushed[0]
print("ep size", ep["size"], "q nan", bool(torch.isnan(ep["q"]).any()), "td", ep["td"][:6].tolist(), "hid absmax", float(ep["hid"].float().abs().max()))
tree = buf.priority_tree.tree()
print("tree nan", bool(torch.isnan(tree).any()), "root", float(tree[0]), "min leaf>0", float(tree[-buf.priority_tree.capacity:][tree[-buf.priority_tree.capacity:] > 0].min()))
batch = buf.sample_batch(B)
names = ["obs", "action", "reward", "done", "steps", "bt_steps", "hidden", "comm", "idx", "weights"]
for n, v in zip(names, batch[:10]):
    vf = v.float()
    print(n, tuple(v.shape), v.dtype, "nan", bool(torch.isnan(vf).any()), "min", float(vf.min()), "max", float(vf.max()))
td, q, qn = lr.compute_td(batch)
print("q", q.view(-1).tolist()); print("qn", qn.view(-1).tolist()); print("td", td.view(-1).tolist())
fin = lambda: all(bool(torch.isfinite(p).all()) for p in lr.model.parameters())
orig = buf.sample_batch
def spy(*a, **k):
    out = orig(*a, **k); buf._last = out; return out
buf.sample_batch = spy
import mapf_rl_amd.model as MM
MM.Network.ENCODE_CHUNK = int(os.environ.get('TCHUNK', 32768))
for k in range(int(os.environ.get('TUPD', 10))):
    t0 = time.time(); out = lr.update(); torch.cuda.synchronize(); print('dt %.3f' % (time.time() - t0))
    torch.cuda.synchronize()
    ok = bool(torch.isfinite(out["q_next"]).all())
    print("update", k, "loss", float(out["loss"]), "qn finite", ok, flush=True)
    if not ok:
        b = buf._last
        nb = b[5] + b[4].view(-1).long()
        tm = lr.tar_model
        with torch.no_grad():
            cnt = 0
            for r in range(10):
                qq = tm.bootstrap(b[0], nb, b[6], b[7]); cnt += int(not bool(torch.isfinite(qq).all()))
            print("  re-run target bootstrap 10x on the same batch: nonfinite runs =", cnt, flush=True)
            with tm._autocast(b[0].device):
                lat = tm.encode(b[0].reshape(-1, 6, 9, 9))
            print("  encoder out finite", bool(torch.isfinite(lat).all()), "absmax", float(lat.float().abs().max()), flush=True)
            lat2 = torch.cat([tm.encode(c) for c in b[0].reshape(-1, 6, 9, 9).split(32768)]) if False else None
            # step through the recurrent part
            B_, T_, N_ = b[0].shape[:3]
            with tm._autocast(b[0].device):
                latv = lat.view(B_, T_, N_, -1); h = b[6].to(latv.dtype)
                for t in range(T_):
                    h = tm.recurrent(latv[:, t].reshape(B_ * N_, -1), h)
                    f1 = bool(torch.isfinite(h).all())
                    h = tm.comm(h.view(B_, N_, 256), b[7][:, t]).reshape(B_ * N_, 256)
                    f2 = bool(torch.isfinite(h).all())
                    if not (f1 and f2):
                        print("  first nonfinite at t", t, "after gru", f1, "after comm", f2, "comm rows sum max", int(b[7][:, t].sum(-1).max()), flush=True)
                        break
        break
