TV=128,1128,64,1064 timeout 600 python tools/tune_bench.py 2>&1 | grep -v amdgpu.ids | tail -4
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof4
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof4 -- python3 $R/tools/profile_update.py > $R/gpurun_out/prof4.log 2>&1; echo prof=$?
cd $R; python tools/summarize_rocprof.py gpurun_out/prof4 "4 learner updates (192x18x40), fused epilogues" | head -30 | cut -c1-150
