# kernel trace + SQ counters of the encoder kernels (separate passes, as gpurun requires)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/enc_trace $R/gpurun_out/enc_pmc $R/gpurun_out/enc_clk
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/enc_trace -- python3 $R/tools/enc_kernels.py > $R/gpurun_out/enc_trace.log 2>&1; echo trace=$?
REPS=1 timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $R/gpurun_out/enc_pmc -- python3 $R/tools/enc_kernels.py > $R/gpurun_out/enc_pmc.log 2>&1; echo pmc=$?
REPS=1 timeout -k 10 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/enc_clk -- python3 $R/tools/enc_kernels.py > $R/gpurun_out/enc_clk.log 2>&1; echo clk=$?
cd $R
python tools/summarize_rocprof.py gpurun_out/enc_trace "encoder kernels" | grep -E "encoder|kernel \|" | cut -c1-170
python - <<'PY'
import csv, glob
from collections import defaultdict
f = glob.glob("gpurun_out/enc_pmc/**/*counter_collection.csv", recursive=True)
agg = defaultdict(lambda: defaultdict(float))
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"]
    if "encoder_" not in k or "pack" in k:
        continue
    name = k.replace("(anonymous namespace)::", "").replace("void ", "")[:48]
    agg[name][r["Counter_Name"]] += float(r["Counter_Value"])
for name, c in agg.items():
    print(name, {k: "%.3g" % v for k, v in c.items()})
    wc = c.get("SQ_WAVE_CYCLES", 0)
    if wc:
        print("   mfma_busy/busy_cycles %.3f  wait_any %.3f  wait_inst %.3f  active %.3f  lds_conflict/lds_active %.3f" % (
            c["SQ_VALU_MFMA_BUSY_CYCLES"] / max(c["SQ_BUSY_CYCLES"], 1), c["SQ_WAIT_ANY"] / wc, c["SQ_WAIT_INST_ANY"] / wc,
            c["SQ_ACTIVE_INST_ANY"] / wc, c["SQ_LDS_BANK_CONFLICT"] / max(c["SQ_LDS_IDX_ACTIVE"], 1)))
PY
python - <<'PY'
# effective shader clock during each encoder kernel: GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / kernel duration
import csv, glob
trace = {}
for f in glob.glob("gpurun_out/enc_clk/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        trace[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"])
for f in glob.glob("gpurun_out/enc_clk/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "encoder_" in r["Kernel_Name"] and "pack" not in r["Kernel_Name"] and r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            ns, name = trace.get(r["Dispatch_Id"], (0, ""))
            if ns:
                print("%-50s %8.3f ms  effective clock %.2f GHz" % (name.replace("(anonymous namespace)::", "").replace("void ", "")[:50], ns / 1e6, float(r["Counter_Value"]) / 8 / ns))
PY
find gpurun_out/enc_trace gpurun_out/enc_pmc gpurun_out/enc_clk -name "*kernel_trace.csv" -delete
