# round 4 closing run: full GPU suite, curriculum iteration (all modes + graph timeline), smoke
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
t0=$(date +%s)
timeout -k 10 700 python -m pytest tests -q -m gpu -x > gpurun_out/r04_final_gputests.log 2>&1; rc=$?; echo gputests=$rc $(( $(date +%s) - t0 ))s
tail -3 gpurun_out/r04_final_gputests.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 200 python tools/curriculum_iter.py 512 200 > gpurun_out/r04_cur_512.txt 2>&1 && echo cur512=0
timeout -k 10 200 python tools/curriculum_iter.py 1024 200 > gpurun_out/r04_cur_1024.txt 2>&1 && echo cur1024=0
cat gpurun_out/r04_cur_512.txt gpurun_out/r04_cur_1024.txt | grep -v amdgpu.ids
cd /tmp
rm -rf $R/gpurun_out/prof_cur
MODES=graph timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_cur -- python3 $R/tools/curriculum_iter.py 512 60 > $R/gpurun_out/prof_cur.log 2>&1; echo prof=$?
cd $R
python tools/update_timeline.py gpurun_out/prof_cur comm_mask_kernel 400 > gpurun_out/r04_curriculum_iteration_timeline.md
find gpurun_out/prof_cur -name "*.csv" -size +1M -delete
cat gpurun_out/r04_curriculum_iteration_timeline.md
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
