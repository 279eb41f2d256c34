#!/usr/bin/env python3
"""Writes tests/golden/fixture_scenarios.npz: the 200 scenarios of each of the reference's shipped evaluation fixtures
test{16,32,64}_40_0.3.pkl (the data `test.py:82-145` evaluates checkpoints on) as bit-packed arrays -- data, not source:

    maps{N}    uint8 [200, 200]     40 x 40 obstacle bits per scenario, np.packbits(..., bitorder="little")
    agents{N}  int8  [200, N, 2]    start cells (row, col)
    goals{N}   int8  [200, N, 2]    goal cells

Read with the restricted unpickler of oracle/ref_harness.py (numpy globals only).  tools/eval_checkpoint.py and
tests/test_eval_gpu.py turn them back into the pkl schema {'maps', 'agents', 'goals'} (mapf_rl_amd.evaluate)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref_harness as rh  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    data = {}
    for nag in (16, 32, 64):
        fx = rh.load_fixture(os.path.join(rh.REFERENCE_DIR, "test%d_40_0.3.pkl" % nag))
        maps = np.stack([np.asarray(m) != 0 for m in fx["maps"]])
        agents = np.stack(fx["agents"])
        goals = np.stack(fx["goals"])
        K, L = maps.shape[0], maps.shape[1]
        assert maps.shape == (K, L, L) and agents.shape == (K, nag, 2) and goals.shape == agents.shape
        assert agents.min() >= 0 and agents.max() < 128 and goals.min() >= 0 and goals.max() < 128
        data["maps%d" % nag] = np.packbits(maps.reshape(K, -1), axis=1, bitorder="little")
        data["agents%d" % nag] = agents.astype(np.int8)
        data["goals%d" % nag] = goals.astype(np.int8)
        data["side%d" % nag] = np.int32(L)
        print("test%d_40_0.3.pkl: %d scenarios, %dx%d, density %.3f" % (nag, K, L, L, maps.mean()))
    np.savez_compressed(os.path.join(OUT, "fixture_scenarios.npz"), **data)


if __name__ == "__main__":
    main()
