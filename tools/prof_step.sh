#!/bin/bash
# per-kernel totals of the timed training steps -> gpurun_out/r03/ (rocprofv3 --kernel-trace --stats)
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r03
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_train -o bench -- python3 $ROOT/bench.py --steps 6 --warmup 2 --sample-batch 0 --no-cpu-baseline > $OUT/bench_train_b128_profiled_run.json 2> $OUT/prof_train.err
find $OUT/prof_train -name "*kernel_stats.csv" -exec cp {} $OUT/bench_train_b128_kernel_stats.csv \;
rm -rf $OUT/prof_train
