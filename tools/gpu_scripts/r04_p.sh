# after the recurrence forward stream: bench line, few-agent update times, curriculum iteration
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python bench.py > gpurun_out/r04_p_bench.json 2> gpurun_out/r04_p_bench.err; echo bench=$?
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_p_bench.json').read().strip().splitlines()[-1])
for k in sorted(d):
    if k.startswith(('learner','actor_loop','train_loop')) or k in ('value','ms_per_step'): print(k, d[k])
PY
timeout -k 10 200 python tools/update_times.py 2>&1 | grep -v amdgpu.ids | tail -4
MODES=graph timeout -k 10 200 python tools/curriculum_iter.py 512 200 2>&1 | grep -v amdgpu.ids
MODES=graph timeout -k 10 200 python tools/curriculum_iter.py 1024 200 2>&1 | grep -v amdgpu.ids
