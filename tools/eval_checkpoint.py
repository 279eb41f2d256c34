#!/usr/bin/env python3
"""A checkpoint on the reference's evaluation fixtures (reference test.py:82-145: finish rate and mean steps over the 200 cases of
test{16,32,64}_40_0.3.pkl), from the committed bit-packed copy of those scenarios (tests/golden/fixture_scenarios.npz).

    python3 tools/eval_checkpoint.py models/12345.pth [more.pth ...] [--random-init]

Prints one line per (checkpoint, fixture): finish rate, mean steps (256 for a case that timed out, as test.py:111-143 counts them)
and the share of agents on their goal at the end."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("checkpoints", nargs="*")
    ap.add_argument("--random-init", action="store_true", help="also evaluate an untrained network (torch.manual_seed(0))")
    ap.add_argument("--fixtures", default=os.path.join(ROOT, "tests", "golden", "fixture_scenarios.npz"))
    ap.add_argument("--agents", type=int, nargs="*", default=[16, 32, 64])
    a = ap.parse_args()
    from mapf_rl_amd.evaluate import evaluate, load_fixture_npz
    from mapf_rl_amd.model import Network

    dev = torch.device("cuda")
    nets = []
    if a.random_init:
        torch.manual_seed(0)
        nets.append(("random-init", Network().to(dev).eval()))
    for ck in a.checkpoints:
        net = Network().to(dev).eval()
        net.load_state_dict(torch.load(ck, map_location=dev))
        nets.append((os.path.basename(ck), net))
    for name, net in nets:
        for n in a.agents:
            tests = load_fixture_npz(a.fixtures, n)
            f, ms, steps, ok, arr = evaluate(net, tests, dev, with_arrivals=True)
            print("%-16s test%d_40_0.3 (200 cases, 40x40, %d agents): finish %.4f   mean steps %.2f   agents on goal at the end %.4f" % (
                name, n, n, f, ms, float(arr.mean())), flush=True)


if __name__ == "__main__":
    main()
