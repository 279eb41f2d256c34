cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 300 python -m pytest tests/test_big_goldens_gpu.py -x -q -s -k update_bf16 2>&1 | grep "GRADERR\|passed\|failed" | sed 's/^.*GRADERR //' > $O/r05_grad_errors_vs_reference.txt; cut -c1-200 $O/r05_grad_errors_vs_reference.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_actor
TACT=40 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_actor -- python3 $R/tools/profile_actor.py > $R/gpurun_out/prof_actor.log 2>&1; echo actor=$?
cd $R
python tools/summarize_rocprof.py gpurun_out/prof_actor actor > gpurun_out/r05_actor_kernel_stats.md
python tools/trace_breakdown.py gpurun_out/prof_actor env_step_kernel 30 > gpurun_out/r05_actor_iteration_breakdown.md
python tools/update_timeline.py gpurun_out/prof_actor env_step_kernel 100 > gpurun_out/r05_actor_iteration_timeline.md
find gpurun_out/prof_actor -name "*kernel_trace.csv" -delete
cat gpurun_out/r05_actor_iteration_timeline.md | cut -c1-100
head -30 gpurun_out/r05_actor_kernel_stats.md | cut -c1-160
tail -2 gpurun_out/prof_actor.log
