# round 4: graph update with the forked backward: tests, timeline, update times; then train.py 5 minutes
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_learner_gpu.py -x -q -m gpu > gpurun_out/r04_e_tests.log 2>&1; rc=$?; echo tests=$rc; tail -5 gpurun_out/r04_e_tests.log
if [ $rc -ne 0 ] || grep -q "Memory access fault" gpurun_out/r04_e_tests.log; then exit 1; fi
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_upd6
NAGENTS=6 MAPLEN=20 TUPD=60 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_upd6 -- python3 $R/tools/profile_update.py > $R/gpurun_out/prof_upd6.log 2>&1; rc=$?; echo prof=$rc
if [ $rc -ne 0 ] || grep -q "Memory access fault" $R/gpurun_out/prof_upd6.log; then tail -5 $R/gpurun_out/prof_upd6.log; exit 1; fi
cd $R
python tools/update_timeline.py gpurun_out/prof_upd6 adam_kernel 400 > gpurun_out/r04_update6_timeline.md
find gpurun_out/prof_upd6 -name "*.csv" -size +1M -delete
head -2 gpurun_out/r04_update6_timeline.md; tail -1 gpurun_out/r04_update6_timeline.md
timeout -k 10 300 python tools/update_times.py 6 20 2048 > gpurun_out/r04_e_update_times_6.log 2>&1; rc=$?; echo ut=$rc; tail -5 gpurun_out/r04_e_update_times_6.log
if [ $rc -ne 0 ] || grep -q "Memory access fault" gpurun_out/r04_e_update_times_6.log; then exit 1; fi
rm -rf models
timeout -k 10 400 python train.py --envs 512 --minutes 5 > gpurun_out/r04_train_curriculum_5min.log 2> gpurun_out/r04_train_5min.err; echo train=$?
tail -14 gpurun_out/r04_train_curriculum_5min.log; tail -3 gpurun_out/r04_train_5min.err
