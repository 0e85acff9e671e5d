#!/bin/bash
# per-kernel totals of the B=128 training step under rocprofv3: bash tools/prof_step_r06.sh <tag> [env assignments...]
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
mkdir -p $ROOT/gpurun_out/r06
cd /tmp && export TMPDIR=/tmp
env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/r06/prof_$TAG -o step -- python3 $ROOT/bench.py --steps 6 --warmup 2 --sample-batch 0 --no-cpu-baseline --no-forward --no-probe > $ROOT/gpurun_out/r06/bench_prof_$TAG.json 2> /dev/null
find $ROOT/gpurun_out/r06/prof_$TAG -name "*kernel_stats.csv" -exec cp {} $ROOT/gpurun_out/r06/kernel_stats_$TAG.csv \;
rm -rf $ROOT/gpurun_out/r06/prof_$TAG
