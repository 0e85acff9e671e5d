#!/bin/bash
# Round-5 same-box A/Bs (run on the GPU box through gpurun; results copied into profiles/r05/).
#   bash tools/ab_r05.sh defer | gnb
cd ${GRAFT_REPO_ROOT:-/root/repo}
step() { env "$@" python3 bench.py --steps 10 --warmup 3 --sample-batch 0 --no-cpu-baseline --no-probe --no-forward $EXTRA 2> /dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('  %.1f img/s %.2f ms/step' % (d['value'], d['ms_per_step']))"; }
case "${1:-defer}" in
defer)      # parameter-gradient reductions: per layer + column-sum passes (round 4's schedule) | per layer | batched (default)
  for r in 1 2 3; do
    echo "per-layer reductions, bias gradients by column-sum passes (PSLD_GN_BWD_COLSUM=0 --per-layer-reductions)"; EXTRA=--per-layer-reductions step PSLD_GN_BWD_COLSUM=0
    echo "per-layer reductions, bias gradients from the GroupNorm backward (--per-layer-reductions)"; EXTRA=--per-layer-reductions step PSLD_X=1
    echo "batched reductions, bias gradients by column-sum passes (PSLD_GN_BWD_COLSUM=0)"; EXTRA= step PSLD_GN_BWD_COLSUM=0
    echo "batched reductions, bias gradients from the GroupNorm backward (default)"; EXTRA= step PSLD_X=1
  done ;;
esac
