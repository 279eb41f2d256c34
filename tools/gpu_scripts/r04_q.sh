# recurrence evidence: A/B against the previous kernel, in-kernel cycle traces (forward, backward), encoder LDS-conflict ablation
cd $GRAFT_REPO_ROOT
timeout -k 10 200 python tools/micro/recur_ab.py run 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_recur_ab.txt; echo ab=$?
for n in 40 6; do timeout -k 10 100 python tools/micro/recur_trace.py run $n 4096 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_recur_trace_$n.txt; echo trace$n=$?; done
for n in 40 6; do timeout -k 10 100 python tools/micro/recur_bwd_trace.py run $n 192 18 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_recur_bwd_trace_$n.txt; echo btrace$n=$?; done
timeout -k 10 200 python tools/micro/enc_ablate.py run 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_encoder_ablate.txt; echo enc=$?
cat gpurun_out/r04_recur_ab.txt | grep -v "max \|bit-ident"
