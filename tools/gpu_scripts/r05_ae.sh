# two ranks of train.py (curriculum) sharing the GPU over gloo: graph-replayed updates with the exchange, asynchronous decisions, pooled level statistics
cd $GRAFT_REPO_ROOT
O=gpurun_out
rm -rf models
MAPF_TRAIN_SHARE_GPU=1 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29519 train.py --envs 256 --minutes 2.5 --interval 15 --learning-starts 20000 --dist-backend gloo > $O/r05_train_2rank_curriculum.log 2> $O/r05_train_2rank_curriculum.err; echo train2=$?
grep "number of updates\|update speed\|buffer update speed\|start training" $O/r05_train_2rank_curriculum.log | tail -8
tail -9 $O/r05_train_2rank_curriculum.log
tail -3 $O/r05_train_2rank_curriculum.err
