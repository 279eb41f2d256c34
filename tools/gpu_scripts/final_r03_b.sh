# end-of-round evidence, part B: config-5 rates, env-step shape sweep, curriculum iteration (launch count), reuse probe, the 2-rank
# bench leg on one GPU, a 5-minute curriculum training run
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03b
rm -rf $O && mkdir -p $O
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O/sweep -- python3 $R/tools/shape_sweep.py > $O/sweep.log 2> $O/sweep.err; echo sweep=$?
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/cur_iter -- python3 $R/tools/curriculum_iter.py 512 60 > $O/curriculum_iter.txt 2>&1; echo cur=$?
cd $R
python3 tools/shape_sweep.py --summarize $O/sweep $O/sweep.log > $O/shape_sweep.md 2>> $O/sweep.err
python tools/summarize_rocprof.py $O/cur_iter curriculum > $O/curriculum_kernel_stats.md
find $O -name "*.csv" -size +1M -delete
cat $O/shape_sweep.md; grep -v amdgpu $O/curriculum_iter.txt; head -6 $O/curriculum_kernel_stats.md
timeout -k 10 300 python tools/curriculum_iter.py 1024 60 2>&1 | grep -v amdgpu >> $O/curriculum_iter.txt; echo cur1024=$?
timeout -k 10 400 python tools/c5_bench.py > $O/c5_rates.txt 2>&1 && timeout -k 10 400 python tools/c5_bench.py --double-q >> $O/c5_rates.txt 2>&1 && timeout -k 10 400 python tools/c5_bench.py 64 40 2048 >> $O/c5_rates.txt 2>&1; echo c5=$?
grep -v amdgpu.ids $O/c5_rates.txt
timeout -k 10 300 python tools/obs_reuse_probe.py --steps 200 2>&1 | grep -v amdgpu > $O/obs_reuse_probe.txt; echo probe=$?; cat $O/obs_reuse_probe.txt
timeout -k 10 300 python tools/actor_times.py 2>&1 | grep -v amdgpu > $O/actor_times.txt; timeout -k 10 300 python tools/actor_times.py --tape 2>&1 | grep -v amdgpu >> $O/actor_times.txt; echo at=$?; grep "reuse=" $O/actor_times.txt
MAPF_BENCH_SHARE_GPU=1 MAPF_BENCH_WATCHDOG=280 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 20 --warmup 5 --dist-backend gloo > $O/bench_2rank.json 2> $O/bench_2rank.err; echo bench2=$?
tail -c 1500 $O/bench_2rank.json
if [ -z "$SKIP_TRAIN" ]; then timeout -k 10 420 python train.py --envs 512 --minutes 5 --interval 20 --learning-starts 20000 2>&1 | grep -v amdgpu > $O/train_curriculum_5min.log; echo train=$?; fi
tail -24 $O/train_curriculum_5min.log
