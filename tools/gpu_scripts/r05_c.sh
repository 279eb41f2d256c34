cd $GRAFT_REPO_ROOT
O=gpurun_out
grep -E "nr_throttled|throttled_usec|nr_periods" /sys/fs/cgroup/cpu.stat
OMP_NUM_THREADS=1 MKL_NUM_THREADS=1 timeout -k 10 240 python tools/two_rank_probe.py --ranks 2 --variant real > $O/r05_probe_2rank_real_omp1.jsonl 2> $O/r05_probe_2rank_real_omp1.err; echo "probe2 omp1 rc=$?"
cat $O/r05_probe_2rank_real_omp1.jsonl
grep -E "nr_throttled|throttled_usec|nr_periods" /sys/fs/cgroup/cpu.stat
timeout -k 10 240 python tools/two_rank_probe.py --ranks 2 --variant real > $O/r05_probe_2rank_real_c.jsonl 2> $O/r05_probe_2rank_real_c.err; echo "probe2 default rc=$?"
cat $O/r05_probe_2rank_real_c.jsonl
grep -E "nr_throttled|throttled_usec|nr_periods" /sys/fs/cgroup/cpu.stat
python - <<'PY'
import torch, os
print("torch threads", torch.get_num_threads(), "interop", torch.get_num_interop_threads(), "affinity", len(os.sched_getaffinity(0)))
print(torch.__config__.parallel_info())
PY
