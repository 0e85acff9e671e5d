#!/bin/bash
# Per-kernel totals of EM sampling steps at B=512 (forward only): bash tools/profile_sample_r03.sh
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r03
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_sample -o sample -- python3 $ROOT/tools/profile_sample.py > $OUT/prof_sample.log 2>&1
find $OUT/prof_sample -name "*kernel_stats.csv" -exec cp {} $OUT/sample_b512_kernel_stats.csv \;
rm -rf $OUT/prof_sample
