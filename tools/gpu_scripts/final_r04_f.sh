# round 4, closing run on the final tree: full GPU suite, smoke, bench line, curriculum iteration (all modes + timeline), config-5 rates, 6-agent update
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
t0=$(date +%s)
timeout -k 10 800 python -m pytest tests -q -m gpu -x > gpurun_out/r04f_gputests.log 2>&1; rc=$?; echo gputests=$rc $(( $(date +%s) - t0 ))s
tail -3 gpurun_out/r04f_gputests.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout -k 10 300 python bench.py > gpurun_out/r04f_bench.json 2> gpurun_out/r04f_bench.err; echo bench=$?
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04f_bench.json').read().strip().splitlines()[-1])
for k in ['value','ms_per_step','learner_ms_per_update','learner_updates_per_sec','actor_loop_ms_per_iter','actor_loop_env_steps_per_sec','actor_loop_tape_policy_ms_per_iter','actor_loop_tape_policy_env_steps_per_sec','actor_loop_every_row_ms_per_iter','train_loop_ms_per_iter','train_loop_updates_per_sec','train_loop_env_steps_per_sec']:
    print(k, d.get(k))
print('roofline', d['roofline']['frac'], d['roofline'].get('frac_out_of_cache'), d['roofline'].get('frac_hbm_proper'), 'cpu', d['cpu_baseline']['value'])
PY
timeout -k 10 200 python tools/curriculum_iter.py 512 200 > gpurun_out/r04f_cur_512.txt 2>&1; echo cur512=$?
timeout -k 10 200 python tools/curriculum_iter.py 1024 200 > gpurun_out/r04f_cur_1024.txt 2>&1; echo cur1024=$?
grep -h "graph=True" gpurun_out/r04f_cur_512.txt gpurun_out/r04f_cur_1024.txt
timeout -k 10 300 python tools/c5_bench.py 2>&1 | tail -1 > gpurun_out/r04f_c5_rates.txt; cat gpurun_out/r04f_c5_rates.txt
timeout -k 10 300 python tools/update_times.py 6 20 2048 2>&1 | grep "prune=True" > gpurun_out/r04f_update6_times.txt; cat gpurun_out/r04f_update6_times.txt
cd /tmp
rm -rf $R/gpurun_out/prof_cur
MODES=graph timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_cur -- python3 $R/tools/curriculum_iter.py 512 60 > $R/gpurun_out/prof_cur.log 2>&1; echo prof_cur=$?
cd $R
python tools/update_timeline.py gpurun_out/prof_cur comm_mask_kernel 400 > gpurun_out/r04f_curriculum_iteration_timeline.md
find gpurun_out/prof_cur -name "*.csv" -size +1M -delete
cat gpurun_out/r04f_curriculum_iteration_timeline.md | cut -c1-100
