# end-of-round evidence (round 6), part A: bench.py kernel stats (the command the bench line comes from), HBM traffic counters of the
# env-step kernel (separate --pmc passes) at 4096 environments, the per-update timelines of the 6-agent (graph-replayed) and 40-agent updates
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06
rm -rf $O && mkdir -p $O
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- python3 $R/bench.py > $O/prof_bench.json 2> $O/prof_bench.err; echo bench=$?
for c in FETCH_SIZE WRITE_SIZE; do
timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_${c}_4096 -- python3 $R/bench.py --no-cpu-baseline --no-dqn --no-out-of-cache --steps 20 --warmup 5 --envs 4096 > $O/pmc_${c}_4096.log 2>&1; echo pmc_${c}=$?
done
NAGENTS=6 MAPLEN=20 NENVS=2048 TUPD=60 MAPF_UPDATE_GRAPH=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_update6 -- python3 $R/tools/profile_update.py > $O/update6.log 2>&1; echo update6=$?
TUPD=8 timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_learner -- python3 $R/tools/profile_update.py > $O/prof_learner.log 2>&1; echo learner=$?
cd $R
python tools/summarize_rocprof.py $O/prof_bench bench env_step_kernel > $O/bench_kernel_stats.md
python tools/update_timeline.py $O/prof_update6 adam_kernel 400 > $O/update6_graph_timeline.md
python tools/summarize_rocprof.py $O/prof_learner learner > $O/learner_kernel_stats.md
python tools/update_timeline.py $O/prof_learner adam_kernel 400 > $O/update40_timeline.md
for c in FETCH_SIZE WRITE_SIZE; do
python tools/pmc_summary.py $O/pmc_${c}_4096 "env_step_kernel<unsigned int, 4, true" > $O/pmc_${c}_4096.txt 2>&1
done
find $O -name "*.csv" -size +1M -delete
cat $O/pmc_FETCH_SIZE_4096.txt $O/pmc_WRITE_SIZE_4096.txt
grep "env_step_kernel" $O/bench_kernel_stats.md | cut -c1-200; tail -3 $O/bench_kernel_stats.md | cut -c1-300
head -1 $O/update6_graph_timeline.md; head -1 $O/update40_timeline.md
tail -c 400 $O/prof_bench.json
