# HBM traffic counters of env_step_kernel (separate --pmc passes), BASELINE size and E = 16384
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03
mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE; do
rm -rf $O/pmc_$c $O/pmc_${c}_16k
timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -- python3 $R/bench.py --no-cpu-baseline --no-dqn --no-out-of-cache --steps 20 --warmup 5 > $O/pmc_$c.log 2>&1; echo pmc_$c=$?
timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_${c}_16k -- python3 $R/bench.py --no-cpu-baseline --no-dqn --no-out-of-cache --steps 20 --warmup 5 --envs 16384 > $O/pmc_${c}_16k.log 2>&1; echo pmc_${c}_16k=$?
done
cd $R
for c in FETCH_SIZE WRITE_SIZE; do
python tools/pmc_summary.py $O/pmc_$c "env_step_kernel<unsigned int, 4, true" > $O/pmc_$c.txt 2>&1
python tools/pmc_summary.py $O/pmc_${c}_16k "env_step_kernel<unsigned int, 4, true" > $O/pmc_${c}_16k.txt 2>&1
done
find $O -name "*.csv" -size +1M -delete
cat $O/pmc_FETCH_SIZE.txt $O/pmc_WRITE_SIZE.txt $O/pmc_FETCH_SIZE_16k.txt $O/pmc_WRITE_SIZE_16k.txt
