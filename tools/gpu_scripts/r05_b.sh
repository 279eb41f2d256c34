cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 240 python tools/two_rank_probe.py --ranks 2 --variant real > $O/r05_probe_2rank_real_b.jsonl 2> $O/r05_probe_2rank_real_b.err; echo "probe2 rc=$?"
cat $O/r05_probe_2rank_real_b.jsonl
nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /sys/fs/cgroup/cpu.stat 2>/dev/null | head -8
