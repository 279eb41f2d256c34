#!/bin/bash
# final tree: two ranks sharing the GPU (gloo, host-staged exchange) and one rank through RCCL
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout -k 10 240 python tools/two_rank_probe.py --ranks 2 --variant real > $O/r05q_probe_2rank.jsonl 2> $O/r05q_probe_2rank.err; echo "probe2 rc=$?"
timeout -k 10 200 python tools/two_rank_probe.py --ranks 1 --variant real --force-dist nccl > $O/r05q_probe_1rank_nccl.jsonl 2> $O/r05q_probe_1rank_nccl.err; echo "probe1 nccl rc=$?"
tail -2 $O/r05q_probe_2rank.jsonl | cut -c1-600; tail -1 $O/r05q_probe_1rank_nccl.jsonl | cut -c1-600
MAPF_BENCH_SHARE_GPU=1 timeout -k 10 400 python bench.py --gpus 2 --dist-backend gloo > $O/r05q_bench_2rank.json 2> $O/r05q_bench_2rank.err; echo "bench2 rc=$?"
python - <<'P'
import json
d=json.loads(open('gpurun_out/r05q_bench_2rank.json').read().strip().splitlines()[-1])
for k in ['value','n_gpus','learner_ms_per_update','actor_loop_ms_per_iter','train_loop_ms_per_iter']: print(k,d.get(k))
P
