ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r04; mkdir -p $OUT
export PMC_OUT=$OUT
PMC_TOOL=bench_limb.py PMC_ARGS="--wgrad" bash $ROOT/tools/pmc_wait.sh > /dev/null 2>&1
python3 $ROOT/tools/pmc_wait_summary.py $OUT 150 dwgrad > $OUT/pmc_wait_dwgrad.md 2>&1
rm -rf $OUT/pmc_wait $OUT/pmc_wait2
PMC_TOOL=bench_wino.py PMC_ARGS="--shapes 256,256,32;512,256,16" bash $ROOT/tools/pmc_wait.sh > /dev/null 2>&1
python3 $ROOT/tools/pmc_wait_summary.py $OUT 100 conv > $OUT/pmc_wait_wino.md 2>&1
rm -rf $OUT/pmc_wait $OUT/pmc_wait2
cat $OUT/pmc_wait_dwgrad.md | head -20; cat $OUT/pmc_wait_wino.md | grep -E "kernel|wino" | head -12
