#!/bin/bash
# 8-wide maps: 64-row tiles with a 16-row LDS pitch (default) vs the round-2 layout; micro-benchmark at three batch sizes
for v in "PSLD_DCONV_W8_PITCH16=1 PSLD_DCONV_W8_MT64=1" "PSLD_DCONV_W8_PITCH16=0 PSLD_DCONV_W8_MT64=1" "PSLD_DCONV_W8_PITCH16=0 PSLD_DCONV_W8_MT64=0"; do
  echo "== $v"
  env $v python3 tools/bench_limb.py --rounds 3 2>&1 | grep "@8 "
  env $v python3 tools/bench_limb.py --rounds 3 --batch 512 2>&1 | grep "@8 "
  env $v python3 tools/bench_limb.py --rounds 3 --batch 16 2>&1 | grep "@8 "
done
