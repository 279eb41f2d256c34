# round 4: the whole GPU suite (timed), the full bench line, the merged-launch sweep
cd $GRAFT_REPO_ROOT
t0=$(date +%s); timeout -k 10 1000 python -m pytest tests -q -m gpu -x > gpurun_out/r04_m_gputests.log 2>&1; rc=$?; echo gputests=$rc seconds=$(( $(date +%s) - t0 )); tail -4 gpurun_out/r04_m_gputests.log
if [ $rc -ne 0 ] || grep -q "Memory access fault" gpurun_out/r04_m_gputests.log; then exit 1; fi
timeout -k 10 500 python bench.py > gpurun_out/r04_m_bench.json 2> gpurun_out/r04_m_bench.err; echo bench=$?
python -c "
import json;r=json.loads(open('gpurun_out/r04_m_bench.json').read().strip().splitlines()[-1])
print(json.dumps(r['roofline']))
for k in r:
    if k.startswith(('learner_','actor_','train_','dqn','encoder')) and not k.endswith(('note','config')): print(k, r[k])
print(r['value'], r['ms_per_step'])"
timeout -k 10 300 python tools/multi_sweep.py 512 2048 8192 32768 > gpurun_out/r04_m_multi_sweep.md 2> gpurun_out/r04_m_multi_sweep.err; echo msweep=$?; cat gpurun_out/r04_m_multi_sweep.md
