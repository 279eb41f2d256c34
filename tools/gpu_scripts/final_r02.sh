# end-of-round evidence: bench.py kernel stats (the same command the bench line comes from), HBM traffic counters of the env-step
# kernel (separate --pmc passes) at the BASELINE size and at E = 16384 (working set beyond the Infinity Cache), actor / learner
# per-iteration breakdowns
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_bench $R/gpurun_out/pmc_* $R/gpurun_out/prof_actor $R/gpurun_out/prof_learner $R/gpurun_out/prof_learner_all $R/gpurun_out/prof_learner128
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench -- python3 $R/bench.py > $R/gpurun_out/prof_bench.json 2> $R/gpurun_out/prof_bench.err; echo bench=$?
for c in FETCH_SIZE WRITE_SIZE; do
timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_$c -- python3 $R/bench.py --no-cpu-baseline --no-dqn --steps 20 --warmup 5 > $R/gpurun_out/pmc_$c.log 2>&1; echo pmc_$c=$?
timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_${c}_16k -- python3 $R/bench.py --no-cpu-baseline --no-dqn --steps 20 --warmup 5 --envs 16384 > $R/gpurun_out/pmc_${c}_16k.log 2>&1; echo pmc_${c}_16k=$?
done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_actor -- python3 $R/tools/profile_actor.py > $R/gpurun_out/prof_actor.log 2>&1; echo actor=$?
TUPD=8 timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_learner -- python3 $R/tools/profile_update.py > $R/gpurun_out/prof_learner.log 2>&1; echo learner=$?
PRUNE=0 TUPD=6 timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_learner_all -- python3 $R/tools/profile_update.py > $R/gpurun_out/prof_learner_all.log 2>&1; echo learner_all=$?
cd $R
python tools/summarize_rocprof.py gpurun_out/prof_bench bench env_step_kernel > gpurun_out/prof_bench.md
python tools/summarize_rocprof.py gpurun_out/prof_actor actor > gpurun_out/prof_actor.md
python tools/summarize_rocprof.py gpurun_out/prof_learner learner > gpurun_out/prof_learner.md
python tools/trace_breakdown.py gpurun_out/prof_actor comm_mask_kernel 24 > gpurun_out/prof_actor_iter.md
python tools/trace_breakdown.py gpurun_out/prof_learner encoder_bwd_kernel 30 > gpurun_out/prof_learner_iter.md
python tools/trace_breakdown.py gpurun_out/prof_learner_all encoder_bwd_kernel 30 > gpurun_out/prof_learner_all_iter.md
python tools/summarize_rocprof.py gpurun_out/prof_learner_all learner_all > gpurun_out/prof_learner_all.md
for c in FETCH_SIZE WRITE_SIZE; do
python tools/pmc_summary.py gpurun_out/pmc_$c "env_step_kernel<unsigned int, 4, true" > gpurun_out/pmc_$c.txt 2>&1
python tools/pmc_summary.py gpurun_out/pmc_${c}_16k "env_step_kernel<unsigned int, 4, true" > gpurun_out/pmc_${c}_16k.txt 2>&1
done
find gpurun_out/prof_bench gpurun_out/prof_actor gpurun_out/prof_learner gpurun_out/prof_learner_all gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE gpurun_out/pmc_FETCH_SIZE_16k gpurun_out/pmc_WRITE_SIZE_16k -name "*.csv" -size +1M -delete
cat gpurun_out/pmc_FETCH_SIZE.txt gpurun_out/pmc_WRITE_SIZE.txt gpurun_out/pmc_FETCH_SIZE_16k.txt gpurun_out/pmc_WRITE_SIZE_16k.txt
head -14 gpurun_out/prof_bench.md; head -20 gpurun_out/prof_learner_iter.md; head -12 gpurun_out/prof_learner_all_iter.md; head -16 gpurun_out/prof_actor_iter.md
tail -c 1200 gpurun_out/prof_bench.json
# config 5's agent count (128): learner breakdown + rates (64x64 / 128 agents / 2048 envs and 40x40 / 64 agents)
cd /tmp
NAGENTS=128 PRUNE=0 TUPD=4 timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_learner128 -- python3 $R/tools/profile_update.py > $R/gpurun_out/prof_learner128.log 2>&1; echo learner128=$?
cd $R
python tools/trace_breakdown.py gpurun_out/prof_learner128 encoder_bwd_kernel 30 > gpurun_out/prof_learner128_iter.md
find gpurun_out/prof_learner128 -name "*.csv" -size +1M -delete
timeout -k 10 400 python tools/c5_bench.py > gpurun_out/c5_rates.txt 2>&1; echo c5=$?
timeout -k 10 400 python tools/c5_bench.py --double-q >> gpurun_out/c5_rates.txt 2>&1; echo c5dq=$?
timeout -k 10 400 python tools/c5_bench.py 64 40 2048 >> gpurun_out/c5_rates.txt 2>&1; echo c5_64=$?
grep -v amdgpu.ids gpurun_out/c5_rates.txt
head -14 gpurun_out/prof_learner128_iter.md
