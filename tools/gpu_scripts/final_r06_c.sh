# end-of-round evidence (round 6), part C: HBM traffic counters of the env-step kernel at the two larger launches (16,384 / 32,768
# environments: the points behind frac_out_of_cache / frac_hbm_proper), separate --pmc passes, as final_r04_a.sh took them
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06c_pmc
rm -rf $O && mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE; do
for e in 16384 32768; do
timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_${c}_$e -- python3 $R/bench.py --no-cpu-baseline --no-dqn --no-out-of-cache --steps 20 --warmup 5 --envs $e > $O/pmc_${c}_$e.log 2>&1; echo pmc_${c}_$e=$?
done
done
cd $R
for c in FETCH_SIZE WRITE_SIZE; do
for e in 16384 32768; do
python tools/pmc_summary.py $O/pmc_${c}_$e "env_step_kernel<unsigned int, 4, true" > $O/pmc_${c}_$e.txt 2>&1
done
done
find $O -name "*.csv" -size +1M -delete
for e in 16384 32768; do echo "== $e environments"; cat $O/pmc_FETCH_SIZE_$e.txt $O/pmc_WRITE_SIZE_$e.txt; done
