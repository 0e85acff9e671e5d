#!/bin/bash
# Timing-only ablations of pw8_kernel per shape (ablation library: make -C tools/abl): bash tools/ab_pw_abl.sh
export PSLD_HIP_LIB=$PWD/tools/abl/libpsld_hip_abl.so
for s in 0 4 6; do
for a in 0 1 2 3 4 7; do
echo "shape $s abl $a: $(PSLD_PW8_ABL=$a python tools/bench_pw.py --only $s 2>/dev/null | head -1)"
done; done
