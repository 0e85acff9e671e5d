#!/bin/bash
# Wait-state counters of the limb convolution kernels (two counter-only passes):  bash tools/pmc_wait.sh
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=${PMC_OUT:-$ROOT/gpurun_out/r02}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_wait -- python3 $ROOT/tools/${PMC_TOOL:-bench_limb.py} --rounds 1 --iters 3 ${PMC_ARGS:-} > /dev/null 2> $OUT/pmc_wait.err
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d $OUT/pmc_wait2 -- python3 $ROOT/tools/${PMC_TOOL:-bench_limb.py} --rounds 1 --iters 3 ${PMC_ARGS:-} > /dev/null 2> $OUT/pmc_wait2.err
tail -n 3 $OUT/pmc_wait.err $OUT/pmc_wait2.err
