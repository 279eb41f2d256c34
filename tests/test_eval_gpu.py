"""GPU: the reference's evaluation (test.py:82-145) on the reference's own fixtures -- the 200 scenarios of each test{16,32,64}_40_0.3.pkl,
committed as bit-packed data (tests/golden/fixture_scenarios.npz, tests/golden/make_fixture_scenarios.py) -- and evidence that training
produces a policy that solves them better than an untrained network.  The full run (curriculum to its stop criterion, 7.5 minutes:
finish 0.835 / 0.595 / 0.065 at 16 / 32 / 64 agents against 0 for random init) is tracked in profiles/r04_eval_after_curriculum.txt."""
import glob
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from tests import helpers as H

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIX = os.path.join(H.GOLDEN, "fixture_scenarios.npz")


def test_fixture_npz_is_the_golden_fixture_data():
    """The committed scenarios are the pkl's: the first cases equal the ones tests/golden/env_fixtures.npz captured with trajectories."""
    from mapf_rl_amd.evaluate import load_fixture_npz

    z = H.load_npz("env_fixtures.npz")
    for nag in (16, 32, 64):
        t = load_fixture_npz(FIX, nag)
        assert len(t["maps"]) == 200 and t["maps"][0].shape == (40, 40) and t["agents"][0].shape == (nag, 2)
        for c in range(3):
            pre = "fix%d_c%d_" % (nag, c)
            assert np.array_equal(t["maps"][c] != 0, z[pre + "map"] != 0)
            assert np.array_equal(t["agents"][c], z[pre + "agents"]) and np.array_equal(t["goals"][c], z[pre + "goals"])


def test_short_training_beats_random_init_on_the_16_agent_fixture(tmp_path):
    """`python train.py` (the reference's curriculum from (1 agent, 10x10)) for 100 seconds, then its checkpoint on
    test16_40_0.3: far more agents reach their goals than under an untrained network (which leaves them where they are)."""
    from mapf_rl_amd.evaluate import evaluate, load_fixture_npz
    from mapf_rl_amd.model import Network

    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "train.py"), "--envs", "512", "--minutes", "1.67", "--interval", "20"], cwd=str(tmp_path),
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    cks = sorted(glob.glob(str(tmp_path / "models" / "*.pth")), key=lambda p: int(os.path.basename(p)[:-4]))
    assert cks, r.stdout[-2000:]
    dev = torch.device("cuda")
    tests = load_fixture_npz(FIX, 16)
    torch.manual_seed(0)
    fresh = Network().to(dev).eval()
    _, _, _, _, arr0 = evaluate(fresh, tests, dev, with_arrivals=True)
    net = Network().to(dev).eval()
    net.load_state_dict(torch.load(cks[-1], map_location=dev))
    f1, steps1, _, _, arr1 = evaluate(net, tests, dev, with_arrivals=True)
    print("random init: %.4f of the agents on their goal at the end; after %s updates: %.4f (finish %.3f, mean steps %.1f)" % (
        arr0.mean(), os.path.basename(cks[-1])[:-4], arr1.mean(), f1, steps1))
    assert arr0.mean() < 0.05
    assert arr1.mean() > arr0.mean() + 0.15
