#!/bin/bash
# which creation order of the role streams keeps the actors' iteration beside the update?
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06n; mkdir -p $O
for ord in "learner_side,capture,actor_stage,actors" "learner_side,capture,actors,actor_stage" "learner_side,actors,capture,actor_stage" "actors,learner_side,capture,actor_stage" "learner_side,-,-,actors,capture,actor_stage" "learner_side,-,-,-,actors,capture,actor_stage" "learner_side,-,-,-,-,actors,capture,actor_stage" "-,learner_side,-,-,actors,capture,actor_stage"; do
MAPF_STREAM_ORDER=$ord MODE=base timeout -k 10 200 python tools/micro/train_loop_overlap.py 1024 100 2>&1 | grep "MODE=" | sed "s/^/ORDER=$ord /" | cut -c1-200 | tee -a $O/order.txt
done
