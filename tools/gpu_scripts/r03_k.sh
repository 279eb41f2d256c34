# round 3 mid-round evidence: learner iteration breakdowns (40 and 128 agents), config-5 rates, bench line
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_learner $R/gpurun_out/prof_learner128
TUPD=8 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_learner -- python3 $R/tools/profile_update.py > $R/gpurun_out/prof_learner.log 2>&1; echo learner=$?
NAGENTS=128 TUPD=6 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_learner128 -- python3 $R/tools/profile_update.py > $R/gpurun_out/prof_learner128.log 2>&1; echo learner128=$?
cd $R
python tools/trace_breakdown.py gpurun_out/prof_learner encoder_bwd_kernel 40 > gpurun_out/prof_learner_iter.md
python tools/trace_breakdown.py gpurun_out/prof_learner128 encoder_bwd_kernel 40 > gpurun_out/prof_learner128_iter.md
find gpurun_out/prof_learner gpurun_out/prof_learner128 -name "*.csv" -size +1M -delete
head -50 gpurun_out/prof_learner_iter.md; head -30 gpurun_out/prof_learner128_iter.md
timeout -k 10 300 python tools/c5_bench.py > gpurun_out/c5_rates.txt 2>&1; echo c5=$?
timeout -k 10 300 python tools/c5_bench.py --double-q >> gpurun_out/c5_rates.txt 2>&1; echo c5dq=$?
timeout -k 10 300 python tools/c5_bench.py 64 40 2048 >> gpurun_out/c5_rates.txt 2>&1; echo c5_64=$?
timeout -k 10 300 python tools/c5_bench.py 6 20 8192 >> gpurun_out/c5_rates.txt 2>&1; echo c6=$?
grep -v amdgpu.ids gpurun_out/c5_rates.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/r03_k_bench.json 2> gpurun_out/r03_k_bench.err; echo bench=$?
python -c "
import json
r=json.loads(open('gpurun_out/r03_k_bench.json').read().strip().splitlines()[-1])
for k in ('value','ms_per_step','learner_ms_per_update','learner_ms_per_update_all_observations','learner_updates_per_sec','actor_loop_ms_per_iter','actor_loop_env_steps_per_sec','train_loop_ms_per_iter','train_loop_updates_per_sec','train_loop_env_steps_per_sec'): print(k, r.get(k))
print(r['roofline'])
"
