# round 4 closing run after the recurrence rework: full GPU suite, bench + kernel stats, learner / actor breakdowns, C5 rates,
# few-agent update, curriculum iteration + timeline, train.py 5 minutes
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
t0=$(date +%s)
timeout -k 10 700 python -m pytest tests -q -m gpu -x > gpurun_out/r04c_gputests.log 2>&1; rc=$?; echo gputests=$rc $(( $(date +%s) - t0 ))s
tail -3 gpurun_out/r04c_gputests.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python bench.py > gpurun_out/r04c_bench.json 2> gpurun_out/r04c_bench.err; echo bench=$?
tail -c 600 gpurun_out/r04c_bench.json
timeout -k 10 300 python tools/c5_bench.py > gpurun_out/r04c_c5_rates.txt 2>&1; echo c5=$?; tail -2 gpurun_out/r04c_c5_rates.txt
timeout -k 10 300 python tools/update_times.py 6 20 2048 > gpurun_out/r04c_update_times_6.log 2>&1; tail -4 gpurun_out/r04c_update_times_6.log
timeout -k 10 200 python tools/curriculum_iter.py 512 200 > gpurun_out/r04c_cur_512.txt 2>&1; echo cur512=$?
timeout -k 10 200 python tools/curriculum_iter.py 1024 200 > gpurun_out/r04c_cur_1024.txt 2>&1; echo cur1024=$?
grep -h "graph=True" gpurun_out/r04c_cur_512.txt gpurun_out/r04c_cur_1024.txt
cd /tmp
rm -rf $R/gpurun_out/prof_cur
MODES=graph timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_cur -- python3 $R/tools/curriculum_iter.py 512 60 > $R/gpurun_out/prof_cur.log 2>&1; echo prof_cur=$?
cd $R
python tools/update_timeline.py gpurun_out/prof_cur comm_mask_kernel 400 > gpurun_out/r04c_curriculum_iteration_timeline.md
find gpurun_out/prof_cur -name "*.csv" -size +1M -delete
head -3 gpurun_out/r04c_curriculum_iteration_timeline.md
