cd $GRAFT_REPO_ROOT
O=gpurun_out/r03b
mkdir -p $O
timeout -k 10 300 python tools/obs_reuse_probe.py --steps 200 2>&1 | grep -v amdgpu > $O/obs_reuse_probe.txt; echo probe=$?; tail -4 $O/obs_reuse_probe.txt
MAPF_BENCH_SHARE_GPU=1 MAPF_BENCH_WATCHDOG=280 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 20 --warmup 5 --dist-backend gloo > $O/bench_2rank.json 2> $O/bench_2rank.err; echo bench2=$?
tail -c 1800 $O/bench_2rank.json
