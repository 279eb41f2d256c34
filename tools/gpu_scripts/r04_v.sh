# the N > 1 leg of bench.py on the one-GPU box: two ranks sharing the card (gloo for the gradient exchange), final code
cd $GRAFT_REPO_ROOT
O=gpurun_out
MAPF_BENCH_SHARE_GPU=1 MAPF_BENCH_WATCHDOG=280 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 20 --warmup 5 --dist-backend gloo > $O/r04_bench_2rank.json 2> $O/r04_bench_2rank.err; echo bench2=$?
tail -c 1500 $O/r04_bench_2rank.json; tail -5 $O/r04_bench_2rank.err
