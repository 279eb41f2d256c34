cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_encoder_gpu.py tests/test_big_goldens_gpu.py tests/test_learner_gpu.py tests/test_entrypoints_gpu.py -q -m gpu 2>&1 | tail -8
timeout -k 10 400 python bench.py --no-cpu-baseline > gpurun_out/r02_h_bench.json 2> gpurun_out/r02_h_bench.err; echo bench=$?
python - <<'PY'
import json
d=json.load(open('gpurun_out/r02_h_bench.json'))
for k in ('value','ms_per_step','learner_updates_per_sec','learner_ms_per_update','actor_loop_env_steps_per_sec','actor_loop_ms_per_iter','train_loop_updates_per_sec','train_loop_ms_per_iter'): print(k, d.get(k))
print(d['roofline'])
PY
timeout -k 10 300 python tools/c5_bench.py 128 64 2048 2>&1 | tail -1
