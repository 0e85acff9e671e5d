#!/bin/bash
# Where does the weight-gradient kernel lose its time?  Timing-only ablations (results wrong by construction):
# 1 = no limb split, 2 = one ds_read_b128 per fragment instead of two transposed reads, 3 = both, 4 = no staging stores, 6 = 2 + 4
for a in 0 1 2 3 4 6; do echo "PSLD_DWGRAD_ABL=$a"; PSLD_HIP_LIB=$PWD/tools/abl/libpsld_hip_abl.so PSLD_DWGRAD_ABL=$a python3 tools/bench_limb.py --wgrad --rounds 3 --iters 5 2>&1 | grep "wgrad" | grep -E "@32|@16" | cut -c1-110; done
