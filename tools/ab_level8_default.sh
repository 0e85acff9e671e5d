cd ${GRAFT_REPO_ROOT:-/root/repo}
step() { env "$@" python3 bench.py --steps 10 --warmup 3 --sample-batch 0 --no-cpu-baseline --no-probe --no-forward --no-config-block $EXTRA 2> /dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('  %.1f img/s %.2f ms/step' % (d['value'], d['ms_per_step']))"; }
for r in 1 2 3; do
  echo "default (8x8 level in Winograd form: split chunks, Winograd-domain weight gradients)"; step PSLD_X=1
  echo "PSLD_WINOGRAD=2 PSLD_WGRAD_WINOGRAD=2 (everything forced)"; step PSLD_WINOGRAD=2 PSLD_WGRAD_WINOGRAD=2
done
echo "CelebA-64"; EXTRA="--config celeba64_sota" step PSLD_X=1; EXTRA="--config celeba64_sota" step PSLD_X=1
echo "B=64"; EXTRA="--batch 64" step PSLD_X=1
echo "B=32"; EXTRA="--batch 32" step PSLD_X=1
echo "B=16"; EXTRA="--batch 16 --steps 30" step PSLD_X=1
