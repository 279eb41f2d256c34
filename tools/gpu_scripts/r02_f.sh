cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/r02_sweep $R/gpurun_out/prof_learner_c5
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r02_sweep -- python3 $R/tools/shape_sweep.py > $R/gpurun_out/r02_sweep.log 2> $R/gpurun_out/r02_sweep.err; echo sweep=$?
NAGENTS=128 TUPD=4 timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_learner_c5 -- python3 $R/tools/profile_update.py > $R/gpurun_out/prof_learner_c5.log 2>&1; echo learner_c5=$?
cd $R
python3 tools/shape_sweep.py --summarize gpurun_out/r02_sweep gpurun_out/r02_sweep.log > gpurun_out/r02_sweep.md 2>> gpurun_out/r02_sweep.err
python tools/summarize_rocprof.py gpurun_out/prof_learner_c5 "learner update, 128 agents" > gpurun_out/prof_learner_c5.md
python tools/trace_breakdown.py gpurun_out/prof_learner_c5 encoder_bwd_kernel 30 > gpurun_out/prof_learner_c5_iter.md
find gpurun_out/r02_sweep gpurun_out/prof_learner_c5 -name "*.csv" -size +1M -delete
cat gpurun_out/r02_sweep.md; head -40 gpurun_out/prof_learner_c5_iter.md
timeout -k 10 900 python -m pytest tests/test_entrypoints_gpu.py tests/test_encoder_gpu.py -q -m gpu 2>&1 | tail -5
