#!/usr/bin/env python3
"""Starts N ranks of a script on THIS box without torch.distributed.run: bench.launch_ranks (fresh children, RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_* set, one failing rank ends the others, the parent never opens the GPU).  For the one-GPU rehearsals of the
multi-rank entry points: a gpurun box lets 6 processes hold the card, and torch.distributed.run's agent process is one of them --
with it only 5 ranks fit, with this launcher 6.

    MAPF_TRAIN_SHARE_GPU=1 python tools/launch_ranks.py 6 train.py --envs 128 --minutes 1 --dist-backend gloo"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

if __name__ == "__main__":
    n, script, argv = int(sys.argv[1]), sys.argv[2], sys.argv[3:]
    if os.environ.get("MAPF_TRAIN_SHARE_GPU") == "1":
        os.environ["MAPF_BENCH_SHARE_GPU"] = "1"  # (the launcher's own "fewer devices than ranks is fine" switch)
    sys.exit(bench.launch_ranks(n, argv, script=os.path.join(ROOT, script) if not os.path.isabs(script) else script,
                                deadline_s=float(os.environ.get("LAUNCH_DEADLINE_S", "540"))))
