#!/bin/bash
# Cost of a K tile by position class (ablation library: every workgroup of the launch takes the SAME position, wrong results):
# 5 = centre (4 + 4 loads per tile and channel quad), 1 = edge (4 + 2), 0 = corner (4 + 1).
cd ${GRAFT_REPO_ROOT:-/root/repo}
make -C tools/abl > /dev/null 2>&1
for pos in -1 5 1 4 0 15; do
  echo "PSLD_WWGRAD_POS=$pos"
  PSLD_HIP_LIB=$PWD/tools/abl/libpsld_hip_abl.so PSLD_WWGRAD_ABL=16 PSLD_WWGRAD_POS=$pos python3 tools/bench_wwgrad.py --rounds 3 --iters 5 --no-ref --shapes 256:0:256:32,256:256:256:32,256:0:256:16 2>&1 | grep "@"
done
