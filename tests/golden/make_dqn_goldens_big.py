#!/usr/bin/env python3
"""Generates tests/golden/dqn_big.npz and dqn_actor.npz from the UNMODIFIED reference (model.py / buffer.py / worker.py of
ZiyuanMa/MAPF_RL, imported through oracle/ref_harness.py) at the BASELINE agent counts.  Build container only; outputs
are data.  (tests/golden/make_dqn_goldens.py holds the small-shape goldens.)

  dqn_big.npz
    step64_*   Network.step (model.py:180-222) on the reference's 64-agent fixture test64_40_0.3.pkl, case 0, 4 steps,
               deterministic weights (tests/helpers.det_state_dict).
    b40_*      a replay-shaped batch B=8, T=18, A=40 of REAL observations / comm masks (reference Environment 32x32 with
               40 agents driven by Network.step), laid out like GlobalBuffer.sample_batch (worker.py:118-162): bt_steps
               burn-in rows + `steps` forward rows, zero rows (observation AND comm mask all False) behind them, initial
               hidden = agent 0's state broadcast to every agent (quirk Q4) or zeros;
               target-style Network.bootstrap (model.py:227-263) Q-values on it, and one Learner.train body
               (worker.py:296-324): td error, priorities, loss, pre-clip gradient norm, and for EVERY parameter a
               fingerprint of its gradient (norm, per-gate block norms, 24 fixed +-1 projections; full tensor when small).
    b6_*       the same at A=6 on 20x20 (the reference's default training shape).
    b128_*     the same at B=2, A=128 on 64x64 (BASELINE config 5's agent count).
  dqn_actor.npz
    Two episodes recorded by the reference's LocalBuffer (buffer.py:108-179) driven exactly like Actor.run
    (worker.py:376-407) with epsilon = 0: (a) policy actions on fixture test16 case 0 until the time limit (quirk Q8: the
    extra model.step on the stale observation), max_steps = 20; (b) a scripted run to `done` on a small scenario.
    Stored: scenario, all agents' actions per step, agent 0's q / reward / hidden, comm masks, observations (bit-packed),
    the finish() tuple's td_errors / done / size, and the top-2 Q gap per agent and step.
"""
import os
import random
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref_harness as rh  # noqa: E402
from tests import helpers as H  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def det_net(ref, seed=1234):
    net = ref.model.Network()
    net.eval()
    sd = H.det_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed=seed)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return net, sd


def f32(x):
    return torch.from_numpy(np.asarray(x).astype(np.float32))


def gen_step64(ref, net, data):
    fx = rh.load_fixture(os.path.join(rh.REFERENCE_DIR, "test64_40_0.3.pkl"))
    env = ref.environment.Environment()
    env.load(fx["maps"][0], fx["agents"][0], fx["goals"][0])
    net.reset()
    obs, pos = env.observe()
    O, P, Q, Hd, CM, A = [], [], [], [], [], []
    for t in range(4):
        actions, q, hidden, comm = net.step(f32(obs), f32(pos))
        O.append(np.packbits(obs.astype(np.uint8).ravel(), bitorder="little"))
        P.append(pos.copy()); Q.append(q.copy()); Hd.append(hidden.copy()); CM.append(comm.copy()); A.append(list(actions))
        (obs, pos), _, _, _ = env.step(actions)
    data["step64_obs_bits"] = np.stack(O)
    data["step64_pos"] = np.stack(P).astype(np.int8)
    data["step64_q"] = np.stack(Q).astype(np.float32)
    data["step64_hidden"] = np.stack(Hd).astype(np.float32)
    data["step64_comm_mask"] = np.packbits(np.stack(CM), axis=-1, bitorder="little")
    data["step64_actions"] = np.array(A, np.int8)
    print("step64 q", data["step64_q"].shape, "comm partners/agent %.2f" % np.stack(CM).sum(-1).mean())


def real_batch(ref, net, B, A, L, seed):
    """Windows shaped like GlobalBuffer.sample_batch from real rollouts (see module docstring)."""
    rng = np.random.RandomState(seed)
    T = 18
    obs = np.zeros((B, T, A, 6, 9, 9), bool)
    comm = np.zeros((B, T, A, A), bool)
    hidden = np.zeros((B, 256), np.float32)
    bt = np.zeros(B, np.int64)
    steps = np.zeros(B, np.int64)
    for b in range(B):
        np.random.seed(seed * 100 + b)
        random.seed(seed * 100 + b)
        env = ref.environment.Environment(num_agents=A, map_length=L)
        net.reset()
        skip = [0, 5, 0, 9, 2, 0, 14, 3][b % 8]
        bt[b] = [16, 16, 3, 16, 9, 1, 16, 12][b % 8]
        steps[b] = [2, 1, 2, 2, 1, 2, 2, 2][b % 8]
        o, p = env.observe()
        for _ in range(skip):
            actions, _, h, _ = net.step(f32(o), f32(p))
            (o, p), _, _, _ = env.step(actions)
            hidden[b] = h[0]
        for t in range(bt[b] + steps[b]):
            actions, _, _, cm = net.step(f32(o), f32(p))
            obs[b, t], comm[b, t] = o.astype(bool), cm
            if rng.random_sample() < 0.3:  # some exploration so that windows are not all alike
                actions[0] = int(rng.randint(5))
            (o, p), _, _, _ = env.step(actions)
    return obs, comm, hidden, bt, steps


def gen_batch(ref, sd, data, tag, B, A, L, seed):
    ref.config.batch_size = B
    learner = ref.worker.Learner(None)
    learner.model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    sd2 = H.det_state_dict({k: tuple(v.shape) for k, v in sd.items()}, seed=777)
    learner.tar_model.load_state_dict({k: torch.from_numpy(v) for k, v in sd2.items()})
    learner.model.eval()
    obs, comm, hid1, bt, steps = real_batch(ref, learner.model, B, A, L, seed)
    learner.model.train()
    rng = np.random.RandomState(seed + 1)
    T = 18
    b_obs = f32(obs)
    b_action = torch.from_numpy(rng.randint(0, 5, size=(B, 1))).long()
    b_reward = f32(rng.choice([-0.075, -0.5, 0.0, 3.0], size=(B, 1)))
    b_done = f32(rng.random_sample((B, 1)) < 0.25)
    b_steps = f32(steps.reshape(B, 1))
    b_bt_steps = torch.from_numpy(bt).long()
    b_hidden = f32(np.repeat(hid1[:, None, :], A, axis=1).reshape(B * A, 256))  # quirk Q4: agent 0's state in every row
    b_comm_mask = torch.from_numpy(comm)
    weights = f32(rng.random_sample((B, 1)) * 0.5 + 0.5)
    # --- body of Learner.train, worker.py:296-324 (GradScaler is inert on CPU) ---
    b_next_bt_steps = torch.LongTensor([(x + y).item() for x, y in zip(b_bt_steps, b_steps)])
    with torch.no_grad():
        q_tar_all = learner.tar_model.bootstrap(b_obs, b_next_bt_steps, b_hidden, b_comm_mask)
        b_q_ = (1 - b_done) * q_tar_all.max(1, keepdim=True)[0]
    q_all = learner.model.bootstrap(b_obs[:, :-ref.config.forward_steps], b_bt_steps, b_hidden, b_comm_mask[:, :-ref.config.forward_steps])
    b_q = q_all.gather(1, b_action)
    td_error = (b_q - (b_reward + (0.99 ** b_steps) * b_q_))
    priorities = td_error.detach().squeeze().abs().clamp(1e-6).cpu().numpy()
    loss = (weights * learner.huber_loss(td_error)).mean()
    learner.optimizer.zero_grad()
    loss.backward()
    grads = {k: p.grad.detach().numpy().copy() for k, p in learner.model.named_parameters()}
    gnorm = torch.nn.utils.clip_grad_norm_(learner.model.parameters(), 40)
    pre = tag + "_"
    data.update({
        pre + "shape": np.array([B, T, A, L]), pre + "obs_bits": np.packbits(obs.ravel(), bitorder="little"),
        pre + "comm_bits": np.packbits(comm.ravel(), bitorder="little"), pre + "hidden0": hid1,
        pre + "action": b_action.numpy(), pre + "reward": b_reward.numpy(), pre + "done": b_done.numpy(),
        pre + "steps": b_steps.numpy(), pre + "bt_steps": b_bt_steps.numpy(), pre + "weights": weights.numpy(),
        pre + "q_target_all": q_tar_all.numpy(), pre + "q_online_all": q_all.detach().numpy(), pre + "q_next": b_q_.numpy(),
        pre + "td": td_error.detach().numpy(), pre + "priorities": priorities, pre + "loss": np.array(loss.item()),
        pre + "grad_norm": np.array(float(gnorm)), pre + "grad_names": np.array(list(grads.keys())),
    })
    for k, g in grads.items():
        fp = H.grad_fingerprint(g)
        data[pre + "gnorm_" + k] = np.array(fp["norm"])
        data[pre + "gproj_" + k] = fp["proj"]
        data[pre + "gblk_" + k] = fp["blocks"]
        if g.size <= 4096:
            data[pre + "gfull_" + k] = g.astype(np.float32)
    print(tag, "loss", loss.item(), "gnorm", float(gnorm), "td", td_error.detach().numpy().ravel().round(3))


def actor_episode(ref, net, env, max_steps, scripted=None):
    """worker.py:376-407 with epsilon = 0 for one episode; `scripted(env, t)` overrides the joint action (all agents)."""
    ref.config.max_steps = max_steps
    obs_pos = env.observe()
    net.reset()
    lb = ref.buffer.LocalBuffer(12, env.num_agents, env.map_size[0], obs_pos[0], size=max_steps)
    rec = dict(actions=[], q0=[], gap=[], hid0=[], comm=[], pos=[obs_pos[1].copy()], reward0=[])
    done = False
    while True:
        actions, q_val, hidden, comm_mask = net.step(f32(obs_pos[0]), f32(obs_pos[1]))
        top2 = np.sort(q_val, axis=1)[:, -2:]
        if scripted is not None:
            actions = scripted(env, len(rec["actions"]))
        next_obs_pos, r, done, _ = env.step(actions)
        lb.add(q_val[0], actions[0], r[0], next_obs_pos[0], hidden[0], comm_mask)
        rec["actions"].append(list(actions)); rec["q0"].append(q_val[0].copy()); rec["gap"].append(top2[:, 1] - top2[:, 0])
        rec["hid0"].append(hidden[0].copy()); rec["comm"].append(comm_mask.copy()); rec["reward0"].append(r[0])
        rec["pos"].append(next_obs_pos[1].copy())
        if done is False and env.steps < max_steps:
            obs_pos = next_obs_pos
        else:
            if done:
                buf = lb.finish()
            else:
                _, q_val, _, comm_mask = net.step(f32(obs_pos[0]), f32(obs_pos[1]))  # quirk Q8: the STALE observation
                rec["last_q0"] = q_val[0].copy()
                buf = lb.finish(q_val[0], comm_mask)
            break
    return buf, rec


def gen_actor(ref, net):
    data = {}
    # (a) time-out episode on the 16-agent fixture, policy actions
    fx = rh.load_fixture(os.path.join(rh.REFERENCE_DIR, "test16_40_0.3.pkl"))
    env = ref.environment.Environment()
    env.load(fx["maps"][0], fx["agents"][0], fx["goals"][0])
    episodes = [("to", env, 20, None)]
    # (b) scripted episode that reaches `done`: every agent follows its navi heuristic (first flagged direction), stays on goal
    np.random.seed(4258)
    random.seed(4258)
    env2 = ref.environment.Environment(num_agents=3, map_length=10)

    def follow(env, t):
        obs, _ = env.observe()
        acts = []
        for i in range(env.num_agents):
            flags = obs[i, 2:6, 4, 4]
            acts.append(int(np.argmax(flags)) + 1 if flags.any() else 0)
        return acts

    episodes.append(("dn", env2, 64, follow))
    for tag, e, ms, scripted in episodes:
        pre = tag + "_"
        data[pre + "map"] = np.asarray(e.map).astype(np.int8)
        data[pre + "agents"] = np.asarray(e.agents_pos).astype(np.int16)
        data[pre + "goals"] = np.asarray(e.goals_pos).astype(np.int16)
        buf, rec = actor_episode(ref, net, e, ms, scripted)
        aid, na, ml, obs_buf, act_buf, rew_buf, hid_buf, td, done, size, comm_buf = buf
        data[pre + "max_steps"] = np.array(ms)
        data[pre + "size"] = np.array(size)
        data[pre + "done"] = np.array(bool(done))
        data[pre + "actions"] = np.array(rec["actions"], np.int8)
        data[pre + "q0"] = np.stack(rec["q0"]).astype(np.float32)
        data[pre + "gap"] = np.stack(rec["gap"]).astype(np.float32)
        data[pre + "hid0"] = np.stack(rec["hid0"]).astype(np.float32)
        data[pre + "pos"] = np.stack(rec["pos"]).astype(np.int16)
        data[pre + "obs_bits"] = np.packbits(obs_buf.ravel(), bitorder="little")
        data[pre + "act_buf"] = act_buf.copy()
        data[pre + "rew_buf"] = rew_buf.astype(np.float32)
        data[pre + "hid_buf0"] = hid_buf[:, 0].astype(np.float32)
        assert all(np.array_equal(hid_buf[:, 0], hid_buf[:, k]) for k in range(na))  # quirk Q4
        data[pre + "comm_buf"] = comm_buf.copy()
        data[pre + "td"] = td.copy()
        if "last_q0" in rec:
            data[pre + "last_q0"] = rec["last_q0"].astype(np.float32)
        print("actor episode", tag, "size", size, "done", done, "td[:4]", td[:4].round(4), "min gap agent0 %.4f" % data[pre + "gap"][:, 0].min())
    np.savez_compressed(os.path.join(OUT, "dqn_actor.npz"), **data)


if __name__ == "__main__":
    ref = rh.load_reference()
    torch.manual_seed(0)
    net, sd = det_net(ref)
    data = {}
    gen_step64(ref, net, data)
    gen_batch(ref, sd, data, "b40", 8, 40, 32, seed=31)
    gen_batch(ref, sd, data, "b6", 8, 6, 20, seed=32)
    gen_batch(ref, sd, data, "b128", 2, 128, 64, seed=33)  # BASELINE config 5's agent count (64x64 map)
    np.savez_compressed(os.path.join(OUT, "dqn_big.npz"), **data)
    print("dqn_big.npz", len(data), "arrays", os.path.getsize(os.path.join(OUT, "dqn_big.npz")) // 1024, "KiB")
    gen_actor(ref, net)
