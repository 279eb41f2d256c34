# SQ counters of env_step_kernel at the small shapes (what bounds it: issue, waits or memory?)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/step_pmc1 $R/gpurun_out/step_pmc2
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $R/gpurun_out/step_pmc1 -- python3 $R/tools/shape_sweep.py 4096,40,16 8192,20,6 4096,32,40 > $R/gpurun_out/step_pmc1.log 2>&1; echo pmc1=$?
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/step_pmc2 -- python3 $R/tools/shape_sweep.py 4096,40,16 8192,20,6 4096,32,40 > $R/gpurun_out/step_pmc2.log 2>&1; echo pmc2=$?
cd $R
python tools/pmc_summary.py gpurun_out/step_pmc1 env_step_kernel | cut -c1-160
python tools/pmc_summary.py gpurun_out/step_pmc2 env_step_kernel | cut -c1-160
tail -3 gpurun_out/step_pmc1.log
