#!/usr/bin/env python3
"""Generates tests/golden/dqn_*.npz from the UNMODIFIED reference (model.py / buffer.py / worker.py of
ZiyuanMa/MAPF_RL, imported through oracle/ref_harness.py).  Build container only; outputs are data.

  dqn_model.npz    G4: Network.step (model.py:180-222) on fixture observations with deterministic weights
                   (tests/helpers.det_state_dict -- a counter-based generator, independent of torch's RNG);
                   G7a: Network.bootstrap (model.py:227-263) on a synthetic batch.
  dqn_replay.npz   G5: SumTree known answers + sampling (buffer.py:16-105); G6: LocalBuffer.finish
                   (buffer.py:153-179), GlobalBuffer.add / sample_batch / update_priorities (worker.py:71-203).
  dqn_update.npz   G7b: one Learner.train body (worker.py:296-324) on a fixed batch: td error, priorities,
                   Huber loss, pre-clip gradient norm, a few parameters after the Adam step.

The reference hard-codes config.batch_size in Network.bootstrap / CommBlock (model.py:128,239,245,255, quirk
Q5); the harness sets the reference's config.batch_size attribute to the golden batch size before calling.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref_harness as rh  # noqa: E402
from tests import helpers as H  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def gen_model(ref):
    torch.manual_seed(0)
    data = {}
    net = ref.model.Network()
    net.eval()
    sd = H.det_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed=1234)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    data["param_names"] = np.array(list(sd.keys()))
    data["param_numel"] = np.array([v.size for v in sd.values()])
    fx = {n: rh.load_fixture(os.path.join(rh.REFERENCE_DIR, "test%d_40_0.3.pkl" % n)) for n in (16, 32)}
    for nag, case, T in ((16, 0, 6), (32, 1, 4)):
        m, a, g = fx[nag]["maps"][case], fx[nag]["agents"][case], fx[nag]["goals"][case]
        env = ref.environment.Environment()
        env.load(m, a, g)
        net.reset()
        pre = "step%d_" % nag
        obs, pos = env.observe()
        O, P, Q, Hd, CM, A = [], [], [], [], [], []
        for t in range(T):
            actions, q, hidden, comm = net.step(torch.from_numpy(obs.astype(np.float32)), torch.from_numpy(pos.astype(np.float32)))
            O.append(np.packbits(obs.astype(np.uint8).ravel(), bitorder="little"))
            P.append(pos.copy()); Q.append(q.copy()); Hd.append(hidden.copy()); CM.append(comm.copy()); A.append(list(actions))
            (obs, pos), _, _, _ = env.step(actions)
        data[pre + "obs_bits"] = np.stack(O)
        data[pre + "pos"] = np.stack(P).astype(np.int8)
        data[pre + "q"] = np.stack(Q).astype(np.float32)
        data[pre + "hidden"] = np.stack(Hd).astype(np.float32)
        data[pre + "comm_mask"] = np.stack(CM)
        data[pre + "actions"] = np.array(A, np.int8)
    # bootstrap on a synthetic batch (B=6, T=18, A=4)
    B, T, A = 6, 18, 4
    ref.config.batch_size = B
    rng = np.random.RandomState(5)
    obs = (rng.random_sample((B, T, A, 6, 9, 9)) < 0.3)
    steps = rng.randint(1, T + 1, size=B)
    hidden = (rng.standard_normal((B * A, 256)) * 0.3).astype(np.float32)
    comm = rng.random_sample((B, T, A, A)) < 0.4
    comm |= np.eye(A, dtype=bool)[None, None]
    comm[0, :, :, :] = np.eye(A, dtype=bool)  # a sample with no communication at all
    with torch.no_grad():
        q = net.bootstrap(torch.from_numpy(obs.astype(np.float32)), torch.from_numpy(steps).long(),
                          torch.from_numpy(hidden), torch.from_numpy(comm))
    data["boot_obs_bits"] = np.packbits(obs.ravel(), bitorder="little")
    data["boot_shape"] = np.array([B, T, A])
    data["boot_steps"] = steps.astype(np.int64)
    data["boot_hidden"] = hidden
    data["boot_comm"] = comm
    data["boot_q"] = q.numpy().astype(np.float32)
    np.savez_compressed(os.path.join(OUT, "dqn_model.npz"), **data)
    print("dqn_model.npz", {k: v.shape for k, v in data.items() if k.startswith("boot") or k.endswith("_q")})
    return net, sd


def synth_episode(rng, actor_id, num_agents, map_len, size, done, LocalBuffer):
    """Drives the reference LocalBuffer (buffer.py:108-179) with synthetic transitions."""
    init_obs = rng.random_sample((num_agents, 6, 9, 9)) < 0.25
    lb = LocalBuffer(actor_id, num_agents, map_len, init_obs)
    for t in range(size):
        q = rng.standard_normal(5).astype(np.float32)
        lb.add(q, int(rng.randint(5)), float(rng.choice([-0.075, -0.5, 0.0, 3.0])),
               rng.random_sample((num_agents, 6, 9, 9)) < 0.25,
               (rng.standard_normal(256) * 0.5).astype(np.float32), rng.random_sample((num_agents, num_agents)) < 0.5)
    if done:
        return lb.finish()
    return lb.finish((rng.standard_normal(5)).astype(np.float32), rng.random_sample((num_agents, num_agents)) < 0.5)


def gen_replay(ref):
    data = {}
    SumTree = ref.buffer.SumTree
    # G5 known answer
    st = SumTree(8)
    st.batch_update(np.arange(8), np.arange(1, 9, dtype=np.float64))
    data["st8_tree"] = st.tree.copy()
    # G5 sampling with recorded uniforms
    st = SumTree(1024)
    rng = np.random.RandomState(3)
    pri = rng.random_sample(1024) ** 2
    pri[rng.random_sample(1024) < 0.3] = 0.0
    st.batch_update(np.arange(1024), pri.copy())
    data["st1024_pri"] = pri
    data["st1024_tree_root"] = np.array(st.tree[0])
    for k, (seed, n) in enumerate(((11, 64), (12, 192), (13, 32))):
        np.random.seed(seed)
        interval = st.tree[0] / n
        u = np.random.uniform(0, interval, n)
        np.random.seed(seed)
        idx, p = st.batch_sample(n)
        data["st1024_s%d_u" % k] = u
        data["st1024_s%d_idx" % k] = idx.astype(np.int64)
        data["st1024_s%d_p" % k] = p
    # partial update then re-check tree
    upd_idx = rng.permutation(1024)[:100]
    upd_p = rng.random_sample(100)
    st.batch_update(upd_idx.copy(), upd_p)
    data["st1024_upd_idx"] = upd_idx.astype(np.int64)
    data["st1024_upd_p"] = upd_p
    data["st1024_tree_after"] = st.tree.copy()

    # G6 GlobalBuffer: capacity 4 episode slots, episodes of varied length, ring wrap-around
    GB = ref.worker.GlobalBuffer
    gb = GB(4)
    rng = np.random.RandomState(21)
    eps = [(0, 3, 10, 5, True), (12, 6, 15, 16, False), (3, 2, 10, 17, True), (11, 1, 10, 256, False),
           (10, 4, 20, 40, True), (1, 6, 40, 1, True)]
    data["gb_num_eps"] = np.array(len(eps))
    for k, (aid, na, ml, size, done) in enumerate(eps):
        ep = synth_episode(rng, aid, na, ml, size, done, ref.buffer.LocalBuffer)
        pre = "gb_ep%d_" % k
        data[pre + "meta"] = np.array([aid, na, ml, size, int(done)])
        data[pre + "obs_bits"] = np.packbits(ep[3].ravel(), bitorder="little")
        data[pre + "act"] = ep[4].copy()
        data[pre + "rew"] = ep[5].astype(np.float32)
        data[pre + "hid"] = ep[6].astype(np.float32)
        data[pre + "td"] = ep[7].copy()
        data[pre + "comm"] = ep[10].copy()
        gb.add([ep])
        if k in (3, 5):  # sample after the ring is full / after wrap-around
            np.random.seed(100 + k)
            interval = gb.priority_tree.tree[0] / 24
            u = np.random.uniform(0, interval, 24)
            np.random.seed(100 + k)
            out = gb.sample_batch(24)
            tag = "gb_s%d_" % k
            data[tag + "u"] = u
            data[tag + "obs_bits"] = np.packbits(out[0].numpy().astype(np.uint8).ravel(), bitorder="little")
            data[tag + "action"] = out[1].numpy()
            data[tag + "reward"] = out[2].numpy().astype(np.float32)
            data[tag + "done"] = out[3].numpy().astype(np.float32)
            data[tag + "steps"] = out[4].numpy().astype(np.float32)
            data[tag + "bt_steps"] = out[5].numpy()
            data[tag + "hidden"] = out[6].numpy().astype(np.float32)
            data[tag + "comm"] = out[7].numpy()
            data[tag + "idxes"] = np.asarray(out[8]).astype(np.int64)
            data[tag + "weights"] = out[9].numpy().astype(np.float32)
            data[tag + "old_ptr"] = np.array(out[10])
            data[tag + "size"] = np.array(gb.size)
            data[tag + "tree_root"] = np.array(gb.priority_tree.tree[0])
            if k == 3:
                # update priorities now (nothing stale) ...
                newp = np.abs(np.random.RandomState(7).standard_normal(24)) + 1e-3
                gb.update_priorities(np.asarray(out[8]).copy(), newp.copy(), out[10])
                data[tag + "newp"] = newp
                data[tag + "leaves_after"] = gb.priority_tree.tree[-gb.priority_tree.capacity:].copy()
                # ... and remember this sample to apply a STALE update after two more episodes were added
                stale = (np.asarray(out[8]).copy(), out[10])
    newp2 = np.abs(np.random.RandomState(8).standard_normal(24)) + 1e-3
    gb.update_priorities(stale[0].copy(), newp2.copy(), stale[1])
    data["gb_stale_newp"] = newp2
    data["gb_stale_leaves_after"] = gb.priority_tree.tree[-gb.priority_tree.capacity:].copy()
    data["gb_final_ptr"] = np.array(gb.ptr)
    np.savez_compressed(os.path.join(OUT, "dqn_replay.npz"), **data)
    print("dqn_replay.npz", len(data), "arrays")


def gen_update(ref, sd):
    """One Learner.train body (worker.py:296-324) in fp32 on CPU with deterministic weights."""
    data = {}
    B, A = 6, 3
    ref.config.batch_size = B
    learner = ref.worker.Learner(None)
    learner.model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    sd2 = H.det_state_dict({k: tuple(v.shape) for k, v in sd.items()}, seed=777)
    learner.tar_model.load_state_dict({k: torch.from_numpy(v) for k, v in sd2.items()})
    rng = np.random.RandomState(9)
    T = 18
    b_obs = torch.from_numpy((rng.random_sample((B, T, A, 6, 9, 9)) < 0.3).astype(np.float32))
    b_action = torch.from_numpy(rng.randint(0, 5, size=(B, 1))).long()
    b_reward = torch.from_numpy(rng.choice([-0.075, -0.5, 0.0, 3.0], size=(B, 1)).astype(np.float32))
    b_done = torch.from_numpy((rng.random_sample((B, 1)) < 0.3).astype(np.float32))
    b_steps = torch.from_numpy(rng.randint(1, 3, size=(B, 1)).astype(np.float32))
    b_bt_steps = torch.from_numpy(rng.randint(1, 17, size=B)).long()
    b_hidden = torch.from_numpy((rng.standard_normal((B * A, 256)) * 0.3).astype(np.float32))
    comm = rng.random_sample((B, T, A, A)) < 0.5
    comm |= np.eye(A, dtype=bool)[None, None]
    b_comm_mask = torch.from_numpy(comm)
    weights = torch.from_numpy(rng.random_sample((B, 1)).astype(np.float32) * 0.5 + 0.5)
    # --- body of Learner.train, worker.py:296-324 (GradScaler is inert on CPU) ---
    b_next_bt_steps = torch.LongTensor([(bt + st).item() for bt, st in zip(b_bt_steps, b_steps)])
    with torch.no_grad():
        b_q_ = (1 - b_done) * learner.tar_model.bootstrap(b_obs, b_next_bt_steps, b_hidden, b_comm_mask).max(1, keepdim=True)[0]
    b_q = learner.model.bootstrap(b_obs[:, :-ref.config.forward_steps], b_bt_steps, b_hidden,
                                  b_comm_mask[:, :-ref.config.forward_steps]).gather(1, b_action)
    td_error = (b_q - (b_reward + (0.99 ** b_steps) * b_q_))
    priorities = td_error.detach().squeeze().abs().clamp(1e-6).cpu().numpy()
    loss = (weights * learner.huber_loss(td_error)).mean()
    learner.optimizer.zero_grad()
    loss.backward()
    gnorm = torch.nn.utils.clip_grad_norm_(learner.model.parameters(), 40)
    learner.optimizer.step()
    learner.scheduler.step()
    data.update(shape=np.array([B, T, A]), obs_bits=np.packbits(b_obs.numpy().astype(np.uint8).ravel(), bitorder="little"),
                action=b_action.numpy(), reward=b_reward.numpy(), done=b_done.numpy(), steps=b_steps.numpy(),
                bt_steps=b_bt_steps.numpy(), hidden=b_hidden.numpy(), comm=comm, weights=weights.numpy(),
                q_next=b_q_.numpy(), q=b_q.detach().numpy(), td=td_error.detach().numpy(), priorities=priorities,
                loss=np.array(loss.item()), grad_norm=np.array(float(gnorm)))
    after = learner.model.state_dict()
    for name in ("adv.bias", "state.weight", "obs_encoder.0.bias", "comm.self_attn.W_O.weight", "recurrent.bias_hh"):
        data["after_" + name] = after[name].numpy().copy()
    np.savez_compressed(os.path.join(OUT, "dqn_update.npz"), **data)
    print("dqn_update.npz loss", loss.item(), "gnorm", float(gnorm), "td", td_error.detach().numpy().ravel())


if __name__ == "__main__":
    ref = rh.load_reference()
    net, sd = gen_model(ref)
    gen_replay(ref)
    gen_update(ref, sd)
