#!/bin/bash
# Round-2 micro-benchmarks only (section 4 of tools/profile_r02.sh)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02
mkdir -p $OUT
cd $ROOT
python3 tools/bench_tile.py > $OUT/tile_kernels.txt 2>&1
python3 tools/bench_limb.py > $OUT/limb_planes_ab.txt 2>&1
python3 tools/bench_limb.py --wgrad --rounds 5 --iters 5 > $OUT/wgrad_xlimb_ab.txt 2>&1
python3 tools/bench_limb.py --batch 16 --rounds 3 > $OUT/limb_planes_b16.txt 2>&1
python3 tools/bench_hbm.py > $OUT/hbm_kernels.txt 2>&1
python3 tools/bench_sample.py > $OUT/sampling.txt 2>&1
python3 tools/graph_midfork.py > $OUT/graph_fork_cost.txt 2>&1
python3 tools/graph_cross.py >> $OUT/graph_fork_cost.txt 2>&1
