cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03
mkdir -p $O; rm -rf $O/prof_learner128_all
NAGENTS=128 PRUNE=0 TUPD=6 timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_learner128_all -- python3 $R/tools/profile_update.py > $O/prof_learner128_all.log 2>&1; echo rc=$?
cd $R
python tools/trace_breakdown.py $O/prof_learner128_all encoder_bwd_kernel 16 > $O/c5_learner_all_iteration_breakdown.md
find $O/prof_learner128_all -name "*.csv" -size +1M -delete
cat $O/c5_learner_all_iteration_breakdown.md
python - <<'PY'
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
os.environ["NAGENTS"] = "128"; os.environ["PRUNE"] = "0"; os.environ["TUPD"] = "3"
import runpy
g = runpy.run_path("tools/profile_update.py")
lr = g["lr"]
for k in range(8):
    s0 = torch.cuda.memory_stats()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    lr.update()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    s1 = torch.cuda.memory_stats()
    print("update %d: %.1f ms  device mallocs +%d  frees +%d  retries +%d  reserved %.1f GB" % (k, dt * 1e3,
        s1["num_device_alloc"] - s0["num_device_alloc"], s1["num_device_free"] - s0["num_device_free"], s1["num_alloc_retries"] - s0["num_alloc_retries"], s1["reserved_bytes.all.current"] / 1e9), flush=True)
PY
