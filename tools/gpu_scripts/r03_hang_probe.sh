# reproduce the intermittent 2-rank (shared GPU, gloo) bench hang and look at the GPU while it hangs
cd $GRAFT_REPO_ROOT
for attempt in 1 2 3 4; do
  MAPF_BENCH_WATCHDOG=80 MAPF_BENCH_SHARE_GPU=1 timeout -k 10 120 python bench.py --gpus 2 --steps 5 --warmup 2 --envs 256 --dist-backend gloo --no-cpu-baseline --dqn-updates 1 --dqn-actor-iters 1 --train-iters 1 > gpurun_out/r03_hang_$attempt.log 2>&1 &
  pid=$!
  for s in $(seq 1 50); do
    sleep 1
    if ! kill -0 $pid 2>/dev/null; then break; fi
  done
  if kill -0 $pid 2>/dev/null; then
    echo "attempt $attempt: still running after 50 s"
    rocm-smi --showuse --showmemuse 2>/dev/null | grep -i "busy\|use" | head -6
    wait $pid
    echo "exit $?"
    grep -n "Timeout\|File" gpurun_out/r03_hang_$attempt.log | head -30
    break
  else
    wait $pid; echo "attempt $attempt: finished rc=$?"
  fi
done
