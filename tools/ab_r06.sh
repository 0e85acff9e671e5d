#!/bin/bash
# Round-6 same-box A/Bs (run on the GPU box through gpurun; results copied into profiles/r06/).
#   bash tools/ab_r06.sh wgrad [celeba]
cd ${GRAFT_REPO_ROOT:-/root/repo}
step() { env "$@" python3 bench.py --steps 10 --warmup 3 --sample-batch 0 --no-cpu-baseline --no-probe --no-forward $EXTRA 2> /dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('  %.1f img/s %.2f ms/step' % (d['value'], d['ms_per_step']))"; }
case "${1:-wgrad}" in
wgrad)      # weight gradients of the 3x3 convolutions: direct limb kernels (round 5) | Winograd domain (default)
  [ "$2" = celeba ] && EXTRA="--config celeba64_sota"
  for r in 1 2 3; do
    echo "direct weight gradients (PSLD_WGRAD_WINOGRAD=0)"; step PSLD_WGRAD_WINOGRAD=0
    echo "Winograd-domain weight gradients (default)"; step PSLD_WGRAD_WINOGRAD=1
  done ;;
level8)     # the 8x8 level in Winograd form (forward / data gradient and, with fp32 activations, the weight gradient)
  for r in 1 2 3; do
    echo "default policies"; step PSLD_X=1
    echo "8x8 level forward / dgrad in Winograd form, its weight gradients direct (PSLD_WINOGRAD=2)"; step PSLD_WINOGRAD=2
    echo "8x8 level forward / dgrad AND weight gradients in Winograd form (PSLD_WINOGRAD=2 PSLD_WGRAD_WINOGRAD=2)"; step PSLD_WINOGRAD=2 PSLD_WGRAD_WINOGRAD=2
  done ;;
esac
