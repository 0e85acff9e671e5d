ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r04; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_step -o step -- python3 $ROOT/bench.py --steps 6 --warmup 2 --sample-batch 0 --no-cpu-baseline --no-forward > $OUT/bench_train_b128_profiled_run.json 2> $OUT/prof_step.err
find $OUT/prof_step -name "*kernel_stats.csv" -exec cp {} $OUT/bench_train_b128_kernel_stats.csv \;
rm -rf $OUT/prof_step
cd $ROOT
python3 bench.py > $OUT/bench_default_run.json 2> $OUT/bench_default_run.err
python3 bench.py --batch 16 --steps 30 --warmup 5 --sample-batch 0 --no-cpu-baseline --no-probe > $OUT/bench_b16_eager.json 2>/dev/null
python3 bench.py --batch 16 --steps 30 --warmup 5 --sample-batch 0 --no-cpu-baseline --no-probe --tape > $OUT/bench_b16_tape.json 2>/dev/null
PSLD_WINOGRAD=0 python3 bench.py --steps 10 --warmup 3 --sample-steps 30 --no-cpu-baseline > $OUT/bench_winograd_off.json 2>/dev/null
PSLD_FUSED_GN=0 python3 bench.py --steps 5 --warmup 2 --sample-steps 100 --no-cpu-baseline > $OUT/bench_fused_gn_off.json 2>/dev/null
python3 bench.py --steps 5 --warmup 2 --sample-steps 100 --no-cpu-baseline > $OUT/bench_fused_gn_on.json 2>/dev/null
PSLD_DIST_BACKEND=gloo PSLD_SHARE_GPU=1 python3 bench.py --gpus 2 --steps 3 --warmup 1 --batch 32 --sample-batch 64 --sample-steps 10 --no-cpu-baseline > $OUT/rehearsal_gloo_2rank.json 2>/dev/null
tail -1 $OUT/bench_default_run.json | cut -c1-200
