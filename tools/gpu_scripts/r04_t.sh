cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_upd40s
MAPF_SIDE_MAX_ROWS=0 TUPD=8 timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_upd40s -- python3 $R/tools/profile_update.py > $R/gpurun_out/prof_upd40s.log 2>&1; echo prof=$?
cd $R
python tools/update_timeline.py gpurun_out/prof_upd40s adam_kernel 400 > gpurun_out/r04_update40_serial_timeline.md
find gpurun_out/prof_upd40s -name "*.csv" -size +1M -delete
head -3 gpurun_out/r04_update40_serial_timeline.md
