#!/bin/bash
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
# 2. HBM traffic of the dominant convolution: separate counter-only passes
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/tools/bench_tile.py --quick --iters 3 > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ROOT/tools/bench_tile.py --quick --iters 3 > /dev/null 2> $OUT/pmc_write.err
python3 $ROOT/tools/pmc_traffic.py $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_traffic.json > $OUT/pmc_traffic.log 2>&1
# 3. MFMA-busy / clock / VALU of the limb kernels (fp32 input vs limb planes)
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc_sq -- python3 $ROOT/tools/bench_limb.py --rounds 1 --iters 3 > /dev/null 2> $OUT/pmc_sq.err
python3 $ROOT/tools/pmc_summary.py $OUT/pmc_sq > $OUT/pmc_tile_kernels.md 2>&1
# 3b. wait-state counters of the limb kernels (two counter-only passes each): matrix-pipe busy against wave residency
bash $ROOT/tools/pmc_wait.sh > /dev/null 2>&1
python3 $ROOT/tools/pmc_wait_summary.py $OUT 250 dconv > $OUT/pmc_wait_dconv.md 2>&1
PMC_ARGS=--wgrad bash $ROOT/tools/pmc_wait.sh > /dev/null 2>&1
python3 $ROOT/tools/pmc_wait_summary.py $OUT 150 dwgrad > $OUT/pmc_wait_dwgrad.md 2>&1
cd /tmp
# 4. micro-benchmarks
cd $ROOT
python3 tools/bench_tile.py > $OUT/tile_kernels.txt 2>&1
python3 tools/bench_limb.py > $OUT/limb_planes_ab.txt 2>&1
python3 tools/bench_hbm.py > $OUT/hbm_kernels.txt 2>&1
python3 tools/bench_sample.py > $OUT/sampling.txt 2>&1
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
ls -la $OUT | head -40
