# greedy (few observations change) fixed-level actor iteration at config 2: kernel breakdown
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_greedy
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_greedy -- python3 $R/tools/micro/actor_host.py > $R/gpurun_out/prof_greedy.log 2>&1; echo prof=$?
cd $R
python tools/update_timeline.py gpurun_out/prof_greedy comm_mask_kernel 60 > gpurun_out/r04_actor_greedy_timeline.md
find gpurun_out/prof_greedy -name "*.csv" -size +1M -delete
cat gpurun_out/r04_actor_greedy_timeline.md | cut -c1-110
grep "wall" gpurun_out/prof_greedy.log
