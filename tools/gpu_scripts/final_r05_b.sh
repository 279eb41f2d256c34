# end-of-round evidence (round 5), part B: full GPU suite, smoke, the bench line, recurrence A/B against round 4's kernels + cycle traces,
# curriculum iteration, 6-agent update, GEMM table
cd $GRAFT_REPO_ROOT
O=gpurun_out
t0=$(date +%s)
timeout -k 10 800 python -m pytest tests -q -m gpu -x > $O/r05f_gputests.log 2>&1; rc=$?; echo gputests=$rc $(( $(date +%s) - t0 ))s
tail -3 $O/r05f_gputests.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout -k 10 400 python bench.py > $O/r05f_bench.json 2> $O/r05f_bench.err; echo bench=$?
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05f_bench.json').read().strip().splitlines()[-1])
for k in ['value','ms_per_step','learner_ms_per_update','learner_updates_per_sec','actor_loop_ms_per_iter','actor_loop_env_steps_per_sec','actor_loop_tape_policy_ms_per_iter','pipeline_env_steps_per_sec','actor_loop_every_row_ms_per_iter','train_loop_ms_per_iter','train_loop_updates_per_sec','train_loop_env_steps_per_sec','dqn_error']:
    print(k, d.get(k))
print('roofline', d['roofline']['frac'], d['roofline'].get('frac_out_of_cache'), d['roofline'].get('frac_hbm_proper'), d['roofline'].get('frac_hbm_proper_2x'), 'cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
print('encoder', {k: v for k, v in d['encoder_roofline'].items() if k != 'clock_note'})
PY
timeout -k 10 300 python tools/micro/recur_multi.py run > $O/r05f_recurrence_ab.txt 2>&1; echo recur_ab=$?; cat $O/r05f_recurrence_ab.txt
python tools/micro/recur_trace.py run 40 4096 > $O/r05f_recur_trace_40.txt 2>&1; python tools/micro/recur_trace.py run 6 192 > $O/r05f_recur_trace_6.txt 2>&1; head -3 $O/r05f_recur_trace_40.txt
timeout -k 10 200 python tools/curriculum_iter.py 512 200 > $O/r05f_cur_512.txt 2>&1; echo cur512=$?
timeout -k 10 200 python tools/curriculum_iter.py 1024 200 > $O/r05f_cur_1024.txt 2>&1; echo cur1024=$?
grep -h "graph=True" $O/r05f_cur_512.txt $O/r05f_cur_1024.txt
timeout -k 10 300 python tools/update_times.py 6 20 2048 2>&1 | grep "prune=True" > $O/r05f_update6_times.txt; cat $O/r05f_update6_times.txt
timeout -k 10 300 python tools/micro/tall_gemm_bench.py > $O/r05f_tall_gemm_bench.txt 2>&1; tail -22 $O/r05f_tall_gemm_bench.txt
