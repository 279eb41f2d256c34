cd $GRAFT_REPO_ROOT
O=gpurun_out
for t in 256 384 512 768; do
echo "== target $t"
MAPF_TALL_WGS=$t timeout -k 10 300 python tools/micro/tall_gemm_bench.py 2>&1 | sed -n 3,12p
done > $O/r05_tall_sweep.txt
cat $O/r05_tall_sweep.txt
