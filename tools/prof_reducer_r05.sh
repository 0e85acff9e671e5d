#!/bin/bash
# per-kernel totals of the B=128 step WITH a BucketReducer attached (1-rank group, collectives replaced by nothing): what the
# reducer's schedule itself costs.  bash tools/prof_reducer_r05.sh <tag> [rccl_occupancy args...]
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
mkdir -p $ROOT/gpurun_out/r05
export TMPDIR=/tmp
python3 -c "import sys; sys.path.insert(0, '$ROOT/tools'); import rccl_occupancy as r; print(r.build_hog())"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/r05/prof_$TAG -o step -- python3 $ROOT/tools/rccl_occupancy.py "$@" --steps 6 --warmup 2 > $ROOT/gpurun_out/r05/bench_prof_$TAG.json 2> /dev/null
find $ROOT/gpurun_out/r05/prof_$TAG -name "*kernel_stats.csv" -exec cp {} $ROOT/gpurun_out/r05/kernel_stats_$TAG.csv \;
rm -rf $ROOT/gpurun_out/r05/prof_$TAG
