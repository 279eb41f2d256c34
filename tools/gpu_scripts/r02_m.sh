cd $GRAFT_REPO_ROOT
mkdir -p /tmp/train_run && cd /tmp/train_run
timeout -k 10 500 python $GRAFT_REPO_ROOT/train.py --envs 512 --minutes 5 --interval 20 --learning-starts 20000 > $GRAFT_REPO_ROOT/gpurun_out/r02_train_cur.log 2>&1; echo train=$?
tail -40 $GRAFT_REPO_ROOT/gpurun_out/r02_train_cur.log
