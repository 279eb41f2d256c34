# merged multi-level launch: kernel durations from a rocprofv3 kernel trace, one size per run
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for e in 512 2048 8192 32768; do
rm -rf $R/gpurun_out/prof_multi_$e
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_multi_$e -- python3 $R/tools/multi_sweep.py $e > $R/gpurun_out/prof_multi_$e.log 2>&1; echo multi_$e=$?
done
cd $R
python3 - <<'PY'
import csv, glob
LEV=[(4,15),(3,20),(2,25),(6,15),(5,20),(1,30),(4,25)]
print("| envs per level | `env_step_multi_kernel` avg us (120 launches) | sum of the 7 per-level `env_step_kernel` averages us | aggregate algorithmic MB | merged TB/s | merged frac of 8 TB/s | per-level frac |")
print("|---|---|---|---|---|---|---|")
for e in (512,2048,8192,32768):
    f=glob.glob("gpurun_out/prof_multi_%d/**/*kernel_stats.csv"%e, recursive=True)[0]
    rows=list(csv.DictReader(open(f)))
    m=[r for r in rows if "env_step_multi_kernel<true>" in r["Name"]]
    per=[r for r in rows if "env_step_kernel<" in r["Name"] and ", true, true," in r["Name"]]
    mu=float(m[0]["AverageNs"])/1e3
    pu=sum(float(r["AverageNs"]) for r in per)/1e3
    alg=sum(L*L+821*N+1 for N,L in LEV)*e
    print("| %d | %.1f | %.1f (%d kernels) | %.1f | %.2f | %.3f | %.3f |" % (e, mu, pu, len(per), alg/1e6, alg/mu/1e6, alg/mu/8e6, alg/pu/8e6))
PY
find gpurun_out/prof_multi_* -name "*.csv" -size +1M -delete
timeout -k 10 300 bash -c 'cd '$R'; t0=$(date +%s); python bench.py > gpurun_out/r04_n_bench.json 2> gpurun_out/r04_n_bench.err; echo bench=$? seconds=$(( $(date +%s) - t0 ))'
python3 -c "
import json;r=json.loads(open('gpurun_out/r04_n_bench.json').read().strip().splitlines()[-1]);print(r['roofline']['frac'], r['encoder_roofline']['frac'], r['learner_ms_per_update'])"
