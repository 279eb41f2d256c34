"""Evaluation harness and scenario-file I/O: the reference's `test.py` (reference test.py:23-145) with the
200 cases of a fixture stepped in lock-step on the GPU instead of one after another on the CPU.

pkl schema (reference test.py:27,76-79 and the shipped test{16,32,64}_40_0.3.pkl):
    {'maps': [K x ndarray (L,L)], 'agents': [K x ndarray (N,2) int64], 'goals': [K x ndarray (N,2) int64]}
(the shipped files carry no 'opt_steps'; it needs the CBS expert of search.py)."""
import os
import pickle

import numpy as np
import torch

from .environment import VecEnvironment, generate_scenarios
from .model import Network

TEST_NUM = 200      # reference test.py:19
MAX_STEPS = 256     # config.max_steps
SAVE_INTERVAL = 2500


class _NumpyOnlyUnpickler(pickle.Unpickler):
    _ALLOWED = {("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"),
                ("numpy", "ndarray"), ("numpy", "dtype")}

    def find_class(self, module, name):
        if (module, name) in self._ALLOWED:
            return super().find_class(module, name)
        raise pickle.UnpicklingError("forbidden global %s.%s in scenario file" % (module, name))


def load_tests(path):
    """Reads a reference-format scenario pkl (numpy arrays only are accepted)."""
    with open(path, "rb") as f:
        tests = _NumpyOnlyUnpickler(f).load()
    assert {"maps", "agents", "goals"} <= set(tests.keys())
    return tests


def load_fixture_npz(path, num_agents):
    """The reference's shipped fixtures as committed bit-packed data (tests/golden/fixture_scenarios.npz, written by
    tests/golden/make_fixture_scenarios.py from test{16,32,64}_40_0.3.pkl) back in the pkl schema `evaluate` takes."""
    z = np.load(path)
    L = int(z["side%d" % num_agents])
    packed = z["maps%d" % num_agents]
    maps = np.unpackbits(packed, axis=1, bitorder="little")[:, :L * L].reshape(-1, L, L)
    return {"maps": [m.astype(np.float32) for m in maps], "agents": [a.astype(np.int64) for a in z["agents%d" % num_agents]],
            "goals": [g.astype(np.int64) for g in z["goals%d" % num_agents]]}


def save_tests(path, maps, agents, goals, extra=None):
    tests = {"maps": [np.asarray(m) for m in maps], "agents": [np.asarray(a, dtype=np.int64) for a in agents],
             "goals": [np.asarray(g, dtype=np.int64) for g in goals]}
    if extra:
        tests.update(extra)
    with open(path, "wb") as f:
        pickle.dump(tests, f, protocol=4)
    return tests


def create_test(agent_range, map_range, test_num=TEST_NUM, density=-1.0, seed=1, path=None, with_opt_steps=False,
                time_limit=5.0):
    """reference test.py:23-79: writes ./test{agents}_{map}.pkl.  with_opt_steps=True labels every scenario with the
    makespan of the expert plan (`opt_steps`, test.py:50-58); a scenario the planner cannot solve within
    `time_limit` seconds is re-drawn, like the reference's `while actions is None: env.reset()` loop."""
    assert isinstance(agent_range, int) and isinstance(map_range, int), "fixed sizes only (the shipped fixtures' case)"
    path = path or "./test{}_{}.pkl".format(agent_range, map_range)
    maps, agents, goals, _ = generate_scenarios(test_num, map_range, agent_range, density, seed)
    extra = None
    if with_opt_steps:
        from .search import plan

        opt = []
        redraw_seed = seed * 100003
        for k in range(test_num):
            res = plan(maps[k], agents[k], goals[k], time_limit)
            while res is None:
                redraw_seed += 1
                m1, a1, g1, _ = generate_scenarios(1, map_range, agent_range, density, redraw_seed)
                maps[k], agents[k], goals[k] = m1[0], a1[0], g1[0]
                res = plan(maps[k], agents[k], goals[k], time_limit)
            opt.append(len(res[0]))
        extra = {"opt_steps": opt, "opt_mean_steps": sum(opt) / len(opt)}  # test.py:58,76
    return save_tests(path, list(maps.astype(np.float32)), list(agents), list(goals), extra)


@torch.no_grad()
def evaluate(network, tests, device=None, max_steps=MAX_STEPS, num_cases=TEST_NUM, with_arrivals=False):
    """One checkpoint over the cases of a fixture (reference test.py:105-143): returns (finish_rate, mean_steps,
    per-case steps, per-case success).  All cases share one shape and are stepped together.
    with_arrivals=True appends the per-case share of agents standing on their goal when the episode ended or timed out (a finer
    signal than the all-or-nothing finish rate; not a reference statistic)."""
    device = torch.device(device) if device is not None else torch.device("cuda")
    K = min(num_cases, len(tests["maps"]))
    maps = np.stack([np.asarray(m) != 0 for m in tests["maps"][:K]]).astype(np.int8)
    agents = np.stack(tests["agents"][:K]).astype(np.int16)
    goals = np.stack(tests["goals"][:K]).astype(np.int16)
    L, N = maps.shape[1], agents.shape[1]
    env = VecEnvironment(K, L, N, device=device)
    env.load(maps, agents, goals)
    obs, pos = env.observe()
    hidden = None
    steps = torch.full((K,), max_steps, dtype=torch.int64, device=device)
    finished = torch.zeros(K, dtype=torch.bool, device=device)
    zeros = torch.zeros((K, N), dtype=torch.int64, device=device)
    for t in range(max_steps):
        actions, _, hidden, _ = network.step_batch(obs, pos, hidden)
        # an environment whose episode ended is frozen (the reference leaves its loop, test.py:111)
        actions = torch.where(finished[:, None], zeros, actions)
        obs, pos, _, done, _ = env.step(actions.to(torch.int8).contiguous())
        newly = (done != 0) & ~finished
        steps = torch.where(newly, torch.full_like(steps, t + 1), steps)
        finished |= newly
        if bool(finished.all()):
            break
    env.check_status()
    at_goal = (env.agents_pos() == env.goals_pos()).all(-1)
    ok = at_goal.all(-1)   # test.py:130
    out = (float(ok.float().mean()), float(steps.float().mean()), steps.cpu().numpy(), ok.cpu().numpy())
    return out + (at_goal.float().mean(-1).cpu().numpy(),) if with_arrivals else out


def test_model(test_case="test32_40_0.3.pkl", model_dir="./models", start=190000, device=None):
    """reference test.py:82-145: every checkpoint from `start` downward by 2500 while the file exists."""
    device = torch.device(device) if device is not None else torch.device("cuda")
    network = Network().to(device).eval()
    tests = load_tests(test_case)
    results = []
    model_name = start
    while os.path.exists(os.path.join(model_dir, "{}.pth".format(model_name))):
        sd = torch.load(os.path.join(model_dir, "{}.pth".format(model_name)), map_location=device)
        network.load_state_dict(sd)
        f_rate, mean_steps, _, _ = evaluate(network, tests, device)
        print("--------------{}---------------".format(model_name))
        print("finish: %.4f" % f_rate)
        print("mean steps: %.2f" % mean_steps)
        results.append((model_name, f_rate, mean_steps))
        model_name -= SAVE_INTERVAL
    return results
