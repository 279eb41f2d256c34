cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -q -m gpu -x > gpurun_out/r03_g_tests.log 2>&1; echo tests=$?; tail -5 gpurun_out/r03_g_tests.log
python - <<'PY'
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
os.environ["NAGENTS"] = "128"; os.environ["PRUNE"] = "0"; os.environ["TUPD"] = "3"
import runpy
g = runpy.run_path("tools/profile_update.py")
lr = g["lr"]
from mapf_rl_amd.update import FusedUpdate
FusedUpdate.DEDUP = True   # varying row counts: the allocator's hard case
for k in range(14):
    s0 = torch.cuda.memory_stats()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    lr.update()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    s1 = torch.cuda.memory_stats()
    print("update %d: %.1f ms  device mallocs +%d  reserved %.1f GB" % (k, dt * 1e3, s1["num_device_alloc"] - s0["num_device_alloc"], s1["reserved_bytes.all.current"] / 1e9), flush=True)
PY
timeout -k 10 300 python tools/update_times.py 2>&1 | grep -v amdgpu | tail -2
