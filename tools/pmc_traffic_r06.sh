#!/bin/bash
# HBM traffic of the dominant launch (two counter-only passes) -> gpurun_out/r06/pmc_traffic.json; run through gpurun, then copy to profiles/r06/
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/r06; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/tools/bench_wino.py --rounds 1 --iters 3 --shapes 256,256,32 > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ROOT/tools/bench_wino.py --rounds 1 --iters 3 --shapes 256,256,32 > /dev/null 2> $OUT/pmc_write.err
python3 $ROOT/tools/pmc_traffic.py $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_traffic.json > $OUT/pmc_traffic.log 2>&1
rm -rf $OUT/pmc_fetch $OUT/pmc_write
cat $OUT/pmc_traffic.json | head -c 600
