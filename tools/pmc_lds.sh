#!/bin/bash
# LDS bank conflicts / MFMA busy / VALU per kernel:  PMC_TOOL=bench_wino.py PMC_ARGS="--shapes 256,256,32" bash tools/pmc_lds.sh
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=${PMC_OUT:-$ROOT/gpurun_out/r03}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc_sq -- python3 $ROOT/tools/${PMC_TOOL:-bench_limb.py} --rounds 1 --iters 3 ${PMC_ARGS:-} > /dev/null 2> $OUT/pmc_sq.err
python3 $ROOT/tools/pmc_summary.py $OUT/pmc_sq > $OUT/${PMC_NAME:-pmc_tile_kernels}.md 2>&1
rm -rf $OUT/pmc_sq
