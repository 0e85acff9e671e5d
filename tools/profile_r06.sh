#!/bin/bash
# Round-6 measurement pass (run on the GPU box through gpurun):  bash tools/profile_r06.sh [quick | hbm]
#   quick: without the side bench lines;  hbm: only the in-situ HBM records (stale after an executor change) + the default bench line
# Everything lands under gpurun_out/r06/; the files that are cited are then copied into profiles/r06/.
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06
P=$ROOT/profiles/r06
mkdir -p $OUT $P
cd $ROOT
MODE=${1:-}
if [ "$MODE" != "hbm" ]; then
# 1. the -m gpu suite on this build
python -m pytest tests -m gpu -q > $OUT/gpu_tests.log 2>&1
grep -E "passed|failed" $OUT/gpu_tests.log | tail -1
# 2. HBM traffic of the dominant launch (two counter-only passes; pmc_traffic.json carries the hash of the kernel sources)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/tools/bench_wino.py --rounds 1 --iters 3 --shapes 256,256,32 > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ROOT/tools/bench_wino.py --rounds 1 --iters 3 --shapes 256,256,32 > /dev/null 2> $OUT/pmc_write.err
python3 $ROOT/tools/pmc_traffic.py $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_traffic.json > $OUT/pmc_traffic.log 2>&1
rm -rf $OUT/pmc_fetch $OUT/pmc_write
cp $OUT/pmc_traffic.json $P/pmc_traffic.json
fi
cd /tmp && export TMPDIR=/tmp
# 3. in-situ HBM rate of the bandwidth-bound kernels: the training step, and the eval forward alone (north_star's figure)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/hbm_prof -- python3 $ROOT/tools/hbm_in_situ.py run $OUT/hbm_bytes.json > $OUT/hbm_run.log 2>&1
python3 $ROOT/tools/hbm_in_situ.py join $OUT/hbm_prof $OUT/hbm_bytes.json $OUT/hbm_in_situ > $OUT/hbm_join.log 2>&1
find $OUT/hbm_prof -name "*kernel_stats.csv" -exec cp {} $OUT/hbm_step_kernel_stats.csv \;
rm -rf $OUT/hbm_prof
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/hbm_prof_f -- python3 $ROOT/tools/hbm_in_situ.py run-forward $OUT/hbm_bytes_forward.json > $OUT/hbm_run_forward.log 2>&1
python3 $ROOT/tools/hbm_in_situ.py join $OUT/hbm_prof_f $OUT/hbm_bytes_forward.json $OUT/hbm_in_situ_forward > $OUT/hbm_join_forward.log 2>&1
find $OUT/hbm_prof_f -name "*kernel_stats.csv" -exec cp {} $OUT/hbm_forward_kernel_stats.csv \;
rm -rf $OUT/hbm_prof_f
cp $OUT/hbm_in_situ.json $OUT/hbm_in_situ.md $OUT/hbm_in_situ_forward.json $OUT/hbm_in_situ_forward.md $OUT/hbm_step_kernel_stats.csv $OUT/hbm_forward_kernel_stats.csv $P/ 2>/dev/null
if [ "$MODE" != "hbm" ]; then
# 4. per-kernel totals of the bench's training steps (headline pass + probed pass: 2 + 6 + 6 steps) and of the sampling forward
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_step -o step -- python3 $ROOT/bench.py --steps 6 --warmup 2 --sample-batch 0 --no-cpu-baseline --no-forward --no-config-block > $OUT/bench_train_b128_profiled_run.json 2> $OUT/prof_step.err
find $OUT/prof_step -name "*kernel_stats.csv" -exec cp {} $OUT/bench_train_b128_kernel_stats.csv \;
rm -rf $OUT/prof_step
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_sample -o sample -- python3 $ROOT/tools/profile_sample.py > $OUT/prof_sample.log 2>&1
find $OUT/prof_sample -name "*kernel_stats.csv" -exec cp {} $OUT/sample_b512_kernel_stats.csv \;
rm -rf $OUT/prof_sample
cp $OUT/bench_train_b128_kernel_stats.csv $OUT/bench_train_b128_profiled_run.json $OUT/sample_b512_kernel_stats.csv $OUT/gpu_tests.log $P/ 2>/dev/null
fi
cd $ROOT
# 5. the bench lines (the default one last: it is the one the driver reproduces)
if [ "$MODE" != "quick" ] && [ "$MODE" != "hbm" ]; then
python3 bench.py --batch 16 --steps 30 --warmup 5 --sample-batch 0 --no-cpu-baseline --no-probe > $OUT/bench_b16_eager.json 2>/dev/null
python3 bench.py --batch 64 --steps 20 --warmup 5 --sample-batch 0 --no-cpu-baseline --no-probe > $OUT/bench_b64_eager.json 2>/dev/null
python3 bench.py --config celeba64_sota --steps 10 --warmup 3 --sample-batch 0 --no-cpu-baseline > $OUT/bench_celeba64.json 2>/dev/null
PSLD_FORCE_PG=1 python3 bench.py --steps 10 --warmup 3 --sample-batch 0 --no-cpu-baseline > $OUT/bench_rccl_1rank.json 2>/dev/null
cp $OUT/bench_b16_eager.json $OUT/bench_b64_eager.json $OUT/bench_celeba64.json $OUT/bench_rccl_1rank.json $P/ 2>/dev/null
fi
python3 bench.py > $OUT/bench_default_run.json 2> $OUT/bench_default_run.err
cp $OUT/bench_default_run.json $P/
tail -1 $OUT/bench_default_run.json | cut -c1-400
ls $OUT
