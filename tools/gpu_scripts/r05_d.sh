cd $GRAFT_REPO_ROOT
O=gpurun_out
for i in 1 2; do
timeout -k 10 240 python tools/two_rank_probe.py --ranks 2 --variant real > $O/r05_probe_2rank_staged_$i.jsonl 2> $O/r05_probe_2rank_staged_$i.err; echo "probe2 rc=$?"
cat $O/r05_probe_2rank_staged_$i.jsonl
done
timeout -k 10 900 python -m pytest tests/test_learner_gpu.py tests/test_entrypoints_gpu.py -x -q > $O/r05_d_tests.log 2>&1; echo "tests rc=$?"
tail -15 $O/r05_d_tests.log
