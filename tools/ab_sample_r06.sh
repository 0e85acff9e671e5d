#!/bin/bash
# EM step time at B=512 (and B=16 / 64) under the inference-path switches: GroupNorm fused into the Winograd staging on maps >= 32x32
# (default) / every shape / off; eager vs hipGraph replay of the forward.
cd ${GRAFT_REPO_ROOT:-/root/repo}
for r in 1 2; do
for sw in PSLD_FUSED_GN=1 PSLD_FUSED_GN=2 PSLD_FUSED_GN=0; do
  echo "== $sw"
  env $sw ONLY512=1 python3 tools/bench_sample.py 2>&1 | grep "B="
done
done
