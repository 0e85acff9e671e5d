#!/bin/bash
# Round-3 counter passes for the Winograd kernel (run on the GPU box):  bash tools/pmc_r03.sh
# wait-state table + HBM traffic of the 256->256 @32x32 B=128 launch -> gpurun_out/r03/
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r03
mkdir -p $OUT
export PMC_OUT=$OUT PMC_TOOL=bench_wino.py PMC_ARGS="--shapes 256,256,32;512,256,16"
bash $ROOT/tools/pmc_wait.sh > /dev/null 2>&1
python3 $ROOT/tools/pmc_wait_summary.py $OUT 100 conv > $OUT/pmc_wait_wino.md 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/tools/bench_wino.py --rounds 1 --iters 3 --shapes 256,256,32 > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ROOT/tools/bench_wino.py --rounds 1 --iters 3 --shapes 256,256,32 > /dev/null 2> $OUT/pmc_write.err
python3 $ROOT/tools/pmc_traffic.py $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_traffic.json > $OUT/pmc_traffic.log 2>&1
rm -rf $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_wait $OUT/pmc_wait2
