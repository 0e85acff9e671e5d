#!/bin/bash
# The A/B switches of the PRODUCT library and executor must all stay correct: kernel + model + block parity tests under each.
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r04
for sw in PSLD_PW8=0 PSLD_FUSED_ATTN=3 PSLD_FUSED_ATTN=0 PSLD_WINOGRAD=0 PSLD_WINOGRAD=2 PSLD_FUSED_GN=0 PSLD_FUSED_GN=2 PSLD_FIR_QUAD=0 PSLD_GN_BWD_FUSED=0 PSLD_GN_BWD_PIPE=0 PSLD_GN_BWD_COLSUM=0 PSLD_DCONV_MT64=0 PSLD_DCONV_N32=0 PSLD_FUSE_GN_BWD=1 PSLD_LIMB_PLANES=0 PSLD_LP_SINGLE_BUFFER=1 PSLD_OVERLAP_WGRAD=0 PSLD_OVERLAP_WGRAD=1 PSLD_SIDE_GROUP=1; do
  echo "== $sw"
  env $sw python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py tests/test_blocks_gpu.py -q -x 2>&1 | grep -E "passed|failed|Error" | tail -3
done | tee gpurun_out/r04/test_switches.log
