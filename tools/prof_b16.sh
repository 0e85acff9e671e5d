#!/bin/bash
# B=16 training step from the launch tape: host vs GPU time, then per-kernel totals (rocprofv3 --kernel-trace --stats)
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r03
mkdir -p $OUT
cd $ROOT
for m in "" --graphs --tape; do python3 tools/host_vs_gpu.py --batch 16 $m 2>/dev/null | tail -1; done | tee $OUT/host_vs_gpu_b16.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_b16 -o bench -- python3 $ROOT/bench.py --batch 16 --tape --steps 20 --warmup 5 --sample-batch 0 --no-cpu-baseline --no-probe > $OUT/bench_b16_tape_profiled.json 2> $OUT/prof_b16.err
find $OUT/prof_b16 -name "*kernel_stats.csv" -exec cp {} $OUT/bench_b16_tape_kernel_stats.csv \;
rm -rf $OUT/prof_b16
tail -1 $OUT/bench_b16_tape_profiled.json | cut -c1-300
