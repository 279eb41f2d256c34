"""Entry point compatible with the reference's `python3 test.py` (reference test.py:207-213): evaluates the
checkpoints in ./models on a scenario fixture.  The work runs through mapf_rl_amd (GPU, 200 cases in lock-step).

    python3 test.py [--test-case test32_40_0.3.pkl] [--model-dir ./models] [--start 190000]
    python3 test.py --create 32 40     # writes ./test32_40.pkl incl. opt_steps from the CBS expert (reference create_test)
"""
import argparse
import random

import numpy as np
import torch

torch.manual_seed(1)
np.random.seed(1)
random.seed(1)
test_num = 200


def create_test(agent_range, map_range):
    from mapf_rl_amd.evaluate import create_test as _create

    return _create(agent_range, map_range, test_num, with_opt_steps=True)


def test_model(test_case="test32_40_0.3.pkl", model_dir="./models", start=190000):
    from mapf_rl_amd.evaluate import test_model as _test

    return _test(test_case, model_dir, start)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--test-case", default="test32_40_0.3.pkl")
    ap.add_argument("--model-dir", default="./models")
    ap.add_argument("--start", type=int, default=190000)
    ap.add_argument("--create", nargs=2, type=int, metavar=("AGENTS", "MAP"))
    a = ap.parse_args()
    if a.create:
        create_test(a.create[0], a.create[1])
    else:
        test_model(a.test_case, a.model_dir, a.start)
