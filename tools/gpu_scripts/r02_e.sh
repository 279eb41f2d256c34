cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/r02_sweep $R/gpurun_out/prof_actor
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r02_sweep -- python3 $R/tools/shape_sweep.py > $R/gpurun_out/r02_sweep.log 2> $R/gpurun_out/r02_sweep.err; echo sweep=$?
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_actor -- python3 $R/tools/profile_actor.py > $R/gpurun_out/prof_actor.log 2>&1; echo actor=$?
cd $R
python3 tools/shape_sweep.py --summarize gpurun_out/r02_sweep gpurun_out/r02_sweep.log > gpurun_out/r02_sweep.md 2>> gpurun_out/r02_sweep.err
python tools/summarize_rocprof.py gpurun_out/prof_actor actor > gpurun_out/prof_actor.md
python tools/trace_breakdown.py gpurun_out/prof_actor comm_mask_kernel 20 > gpurun_out/prof_actor_iter.md
find gpurun_out/r02_sweep gpurun_out/prof_actor -name "*.csv" -size +1M -delete
cat gpurun_out/r02_sweep.log gpurun_out/r02_sweep.md; head -60 gpurun_out/prof_actor_iter.md
timeout -k 10 400 python tools/c5_bench.py 128 64 2048 > gpurun_out/r02_c5.log 2>&1; echo c5=$?; tail -3 gpurun_out/r02_c5.log
timeout -k 10 300 python tools/c5_bench.py 128 64 2048 --double-q >> gpurun_out/r02_c5.log 2>&1; tail -1 gpurun_out/r02_c5.log
timeout -k 10 300 python tools/c5_bench.py 64 40 2048 >> gpurun_out/r02_c5.log 2>&1; tail -1 gpurun_out/r02_c5.log
