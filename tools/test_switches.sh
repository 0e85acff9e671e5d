#!/bin/bash
# The switches of the PRODUCT library and executor (INTEGRATION.md's table) must all stay correct: kernel + model + block
# parity tests under each non-default setting.  PSLD_MATH=f32 and PSLD_AUTOGRAD_PARAMS have dedicated tests in the suite.
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/${ROUND:-r06}
mkdir -p $OUT
for sw in PSLD_WINOGRAD=0 PSLD_WINOGRAD=2 PSLD_WGRAD_WINOGRAD=0 PSLD_WGRAD_WINOGRAD=2 PSLD_FUSED_ATTN=0 PSLD_FUSED_GN=0 PSLD_FUSED_GN=2 PSLD_LIMB_PLANES=0 PSLD_GN_BWD_PIPE=0 PSLD_GN_BWD_COLSUM=0 PSLD_OVERLAP_WGRAD=0 PSLD_OVERLAP_WGRAD=1; do
  echo "== $sw"
  env $sw python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py tests/test_blocks_gpu.py -q -x 2>&1 | grep -E "passed|failed|Error" | tail -3
done | tee $OUT/test_switches.log
