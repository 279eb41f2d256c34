# end-of-round evidence (round 6), part D: the other BASELINE configurations on the final tree -- config 5's shape (64x64, 128 agents,
# 2048 envs) with and without the double-DQN target, and the env kernel's shape sweep (configs 3 / 5, the fixture, curriculum shapes)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06d_cfg; mkdir -p $O
timeout -k 10 300 python tools/c5_bench.py --double-q 2>&1 | tail -1 > $O/c5_rates_double_q.txt; cat $O/c5_rates_double_q.txt
timeout -k 10 300 python tools/c5_bench.py 2>&1 | tail -1 > $O/c5_rates.txt; cat $O/c5_rates.txt
timeout -k 10 400 python tools/shape_sweep.py > $O/shape_sweep.md 2> $O/shape_sweep.err; echo sweep=$?; cat $O/shape_sweep.md | head -40
