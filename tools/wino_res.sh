#!/bin/bash
# register / spill report of the Winograd kernels (cross-compile, no GPU needed)
cd /root/repo/psld_amd/csrc
/opt/rocm/bin/hipcc -O3 -fPIC --offload-arch=gfx950 -std=c++17 -I../../include -I. -Wall -Wno-unused-function -fno-slp-vectorize "$@" \
  -Rpass-analysis=kernel-resource-usage -c conv_wino.hip -o /tmp/conv_wino_res.o 2>&1 | \
  grep -E "Function Name|TotalSGPRs|  VGPRs:|Scratch|Spill" | sed -e 's/.*remark: *//' -e 's/\[-Rpass.*//' | paste - - - - - - | grep "${KERNEL:-wino_conv8}"
