# round 4: GPU-side timeline of one graph-replayed 6-agent update (rocprofv3 kernel trace)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_upd6
NAGENTS=6 MAPLEN=20 TUPD=60 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_upd6 -- python3 $R/tools/profile_update.py > $R/gpurun_out/prof_upd6.log 2>&1; rc=$?; echo prof=$rc
if [ $rc -ne 0 ] || grep -q "Memory access fault" $R/gpurun_out/prof_upd6.log; then tail -5 $R/gpurun_out/prof_upd6.log; exit 1; fi
cd $R
python tools/update_timeline.py gpurun_out/prof_upd6 adam_kernel 400 > gpurun_out/r04_update6_timeline.md
find gpurun_out/prof_upd6 -name "*.csv" -size +1M -delete
head -5 gpurun_out/r04_update6_timeline.md; tail -3 gpurun_out/r04_update6_timeline.md
