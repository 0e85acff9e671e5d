#!/bin/bash
# NOTE (round 5): kept as the record of how profiles of that round were produced; switches it names that lost their A/B
# (PSLD_FUSED_ATTN=3, PSLD_DWGRAD_WS outside the ablation library, --tape, ...) were removed in round 5 - see git history.
# Round-3 measurement pass (run on the GPU box through gpurun):  bash tools/profile_r03.sh
# Everything lands under gpurun_out/r03/; the files that are cited are then copied into profiles/r03/.
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r03
mkdir -p $OUT
# 1. counter passes of the Winograd kernel (wait states, LDS conflicts, HBM traffic) - pmc_traffic.json feeds roofline.traffic
bash $ROOT/tools/pmc_r03.sh
mkdir -p $ROOT/profiles/r03 && cp $OUT/pmc_traffic.json $ROOT/profiles/r03/pmc_traffic.json
PMC_OUT=$OUT PMC_TOOL=bench_wino.py PMC_ARGS="--shapes 256,256,32;256,256,16" PMC_NAME=pmc_lds_wino bash $ROOT/tools/pmc_lds.sh
PMC_OUT=$OUT PMC_TOOL=bench_limb.py PMC_ARGS="--wgrad" bash $ROOT/tools/pmc_wait.sh > /dev/null 2>&1
python3 $ROOT/tools/pmc_wait_summary.py $OUT 150 dwgrad > $OUT/pmc_wait_dwgrad.md 2>&1
rm -rf $OUT/pmc_wait $OUT/pmc_wait2
# 2. per-kernel totals of the training steps and of the sampling forward (rocprofv3 --kernel-trace --stats)
bash $ROOT/tools/prof_step.sh
bash $ROOT/tools/profile_sample_r03.sh
cd $ROOT
# 3. micro-benchmarks
python3 tools/bench_wino.py --rounds 5 > $OUT/wino_kernels.txt 2>&1
python3 tools/bench_tile.py > $OUT/tile_kernels.txt 2>&1
(python3 tools/bench_pw.py; echo "PSLD_PW8=0 (four-wave kernel on every grid)"; PSLD_PW8=0 python3 tools/bench_pw.py) > $OUT/pw_kernels.txt 2>&1
python3 tools/bench_limb.py --wgrad --rounds 5 --iters 5 > $OUT/wgrad_xlimb_ab.txt 2>&1
python3 tools/bench_hbm.py > $OUT/hbm_kernels.txt 2>&1
python3 tools/bench_sample.py > $OUT/sampling.txt 2>&1
# 4. the bench lines
python3 bench.py > $OUT/bench_default_run.json 2> $OUT/bench_default_run.err
PSLD_WINOGRAD=0 python3 bench.py --steps 10 --warmup 3 --sample-steps 30 --no-cpu-baseline > $OUT/bench_winograd_off.json 2>/dev/null
python3 bench.py --batch 16 --steps 30 --warmup 5 --sample-batch 0 --no-cpu-baseline --no-probe > $OUT/bench_b16_eager.json 2>/dev/null
python3 bench.py --batch 16 --steps 30 --warmup 5 --sample-batch 0 --no-cpu-baseline --graphs > $OUT/bench_b16_graph.json 2>/dev/null
python3 bench.py --batch 32 --steps 30 --warmup 5 --sample-batch 0 --no-cpu-baseline --no-probe > $OUT/bench_b32_eager.json 2>/dev/null
python3 bench.py --batch 64 --steps 20 --warmup 5 --sample-batch 0 --no-cpu-baseline --no-probe > $OUT/bench_b64_eager.json 2>/dev/null
python3 bench.py --config celeba64_sota --steps 10 --warmup 3 --sample-batch 0 --no-cpu-baseline > $OUT/bench_celeba64.json 2>/dev/null
PSLD_FORCE_PG=1 python3 bench.py --steps 10 --warmup 3 --sample-batch 0 --no-cpu-baseline > $OUT/bench_rccl_1rank.json 2>/dev/null
PSLD_DIST_BACKEND=gloo PSLD_SHARE_GPU=1 python3 bench.py --gpus 2 --steps 3 --warmup 1 --batch 32 --sample-batch 64 --sample-steps 10 --no-cpu-baseline > $OUT/rehearsal_gloo_2rank.json 2>/dev/null
(python3 tools/host_vs_gpu.py --batch 2; python3 tools/host_vs_gpu.py --batch 16; python3 tools/host_vs_gpu.py --batch 16 --graphs; python3 tools/host_vs_gpu.py --batch 16 --tape) 2>/dev/null | grep batch > $OUT/host_vs_gpu.txt
python3 bench.py --batch 16 --steps 30 --warmup 5 --sample-batch 0 --no-cpu-baseline --no-probe --tape > $OUT/bench_b16_tape.json 2>/dev/null
python3 bench.py --batch 8 --steps 30 --warmup 5 --sample-batch 0 --no-cpu-baseline --no-probe --tape > $OUT/bench_b8_tape.json 2>/dev/null
python3 bench.py --batch 8 --steps 30 --warmup 5 --sample-batch 0 --no-cpu-baseline --no-probe > $OUT/bench_b8_eager.json 2>/dev/null
ls -la $OUT
