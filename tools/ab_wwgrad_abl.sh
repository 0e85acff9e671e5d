#!/bin/bash
# Timing-only ablations of wwgrad_ws_kernel (libpsld_hip_abl.so, wrong results by construction): where the time goes.
cd ${GRAFT_REPO_ROOT:-/root/repo}
make -C tools/abl > /dev/null 2>&1
for abl in ${ABLS:-0 1 2 3 4 8 12 32 64}; do
  echo "PSLD_WWGRAD_ABL=$abl  (1 no split, 2 no global loads, 4 no MFMAs, 8 no fragment reads, 32 consumers idle, 64 producers idle)"
  PSLD_HIP_LIB=$PWD/tools/abl/libpsld_hip_abl.so PSLD_WWGRAD_ABL=$abl python3 tools/bench_wwgrad.py --rounds 3 --iters 5 --no-ref --shapes 256:0:256:32,256:256:256:32,256:0:256:16 2>&1 | grep "@"
done
