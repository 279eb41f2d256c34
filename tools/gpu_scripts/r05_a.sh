# round 5, call A: where the 2-rank shared-GPU train loop goes (tools/two_rank_probe.py) + the new multi-rank graph tests
cd $GRAFT_REPO_ROOT
O=gpurun_out
mkdir -p $O
for v in real none serial; do
  timeout -k 10 240 python tools/two_rank_probe.py --ranks 2 --variant $v > $O/r05_probe_2rank_$v.jsonl 2> $O/r05_probe_2rank_$v.err; echo "probe2 $v rc=$?"
  cat $O/r05_probe_2rank_$v.jsonl
done
timeout -k 10 200 python tools/two_rank_probe.py --ranks 1 --variant real > $O/r05_probe_1rank.jsonl 2> $O/r05_probe_1rank.err; echo "probe1 rc=$?"
cat $O/r05_probe_1rank.jsonl
timeout -k 10 200 python tools/two_rank_probe.py --ranks 1 --variant real --force-dist nccl > $O/r05_probe_1rank_nccl.jsonl 2> $O/r05_probe_1rank_nccl.err; echo "probe1 nccl rc=$?"
cat $O/r05_probe_1rank_nccl.jsonl; tail -3 $O/r05_probe_1rank_nccl.err
timeout -k 10 200 python tools/two_rank_probe.py --ranks 1 --variant real --force-dist gloo > $O/r05_probe_1rank_gloo.jsonl 2> $O/r05_probe_1rank_gloo.err; echo "probe1 gloo rc=$?"
cat $O/r05_probe_1rank_gloo.jsonl
timeout -k 10 600 python -m pytest tests/test_learner_gpu.py -x -q -k "exchange or graph_replayed" > $O/r05_a_tests.log 2>&1; echo "tests rc=$?"
tail -15 $O/r05_a_tests.log
