# instruction-cache behaviour of env_step_kernel at kernel start (is the I$ cold at every launch?)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/step_pmc3
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $R/gpurun_out/step_pmc3 -- python3 $R/tools/shape_sweep.py 4096,32,40 4096,40,16 > $R/gpurun_out/step_pmc3.log 2>&1; echo pmc3=$?
cd $R
python tools/pmc_summary.py gpurun_out/step_pmc3 "env_step_kernel" | grep "true, true" | cut -c40-170
