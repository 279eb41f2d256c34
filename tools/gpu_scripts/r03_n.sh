cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_actor
TACT=8 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_actor -- python3 $R/tools/profile_actor.py > $R/gpurun_out/prof_actor.log 2>&1; echo actor=$?
cd $R
grep "actor loop" gpurun_out/prof_actor.log
python tools/trace_breakdown.py gpurun_out/prof_actor comm_mask_kernel 30 12 > gpurun_out/prof_actor_iter.md
find gpurun_out/prof_actor -name "*.csv" -size +1M -delete
cat gpurun_out/prof_actor_iter.md
timeout -k 10 200 python tools/obs_reuse_probe.py --steps 120 2>&1 | grep -v amdgpu
