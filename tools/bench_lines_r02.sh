#!/bin/bash
# Round-2 bench lines only (sections 1 and 5 of tools/profile_r02.sh; run after profiles/r02/pmc_traffic.json is in place)
# Round-2 measurement pass (run on the GPU box through gpurun):  bash tools/profile_r02.sh
# Everything lands under gpurun_out/r02/; the files that are cited are then copied into profiles/r02/.
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# 1. per-kernel totals of the timed training steps (same command the bench line comes from)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_train -o bench -- python3 $ROOT/bench.py --steps 6 --warmup 2 --sample-batch 0 --no-cpu-baseline > $OUT/bench_train_b128_profiled_run.json 2> $OUT/prof_train.err
find $OUT/prof_train -name "*kernel_stats.csv" -exec cp {} $OUT/bench_train_b128_kernel_stats.csv \;
cd $ROOT
# 5. the bench lines
python3 bench.py > $OUT/bench_default_run.json 2> $OUT/bench_default_run.err
python3 bench.py --batch 16 --steps 30 --warmup 5 --sample-batch 0 --no-cpu-baseline --no-probe > $OUT/bench_b16_eager.json 2>/dev/null
PSLD_OVERLAP_WGRAD=0 python3 bench.py --batch 16 --steps 30 --warmup 5 --sample-batch 0 --no-cpu-baseline --no-probe > $OUT/bench_b16_eager_no_side_stream.json 2>/dev/null
python3 bench.py --batch 16 --steps 30 --warmup 5 --sample-batch 0 --no-cpu-baseline --graphs > $OUT/bench_b16_graph.json 2>/dev/null
python3 bench.py --batch 32 --steps 30 --warmup 5 --sample-batch 0 --no-cpu-baseline --no-probe > $OUT/bench_b32_eager.json 2>/dev/null
(python3 tools/host_vs_gpu.py --batch 2; python3 tools/host_vs_gpu.py --batch 16) 2>/dev/null | grep batch > $OUT/host_vs_gpu.txt
python3 bench.py --config celeba64_sota --steps 10 --warmup 3 --sample-batch 0 --no-cpu-baseline > $OUT/bench_celeba64.json 2>/dev/null
PSLD_FORCE_PG=1 python3 bench.py --steps 10 --warmup 3 --sample-batch 0 --no-cpu-baseline > $OUT/bench_rccl_1rank.json 2>/dev/null
PSLD_DIST_BACKEND=gloo PSLD_SHARE_GPU=1 python3 bench.py --gpus 2 --steps 3 --warmup 1 --batch 32 --sample-batch 0 --no-cpu-baseline > $OUT/rehearsal_gloo_2rank.json 2>/dev/null
