cd ${GRAFT_REPO_ROOT:-/root/repo}
step() { env "$@" python3 bench.py --steps 30 --warmup 5 --sample-batch 0 --no-cpu-baseline --no-probe --no-forward --no-config-block $EXTRA 2> /dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('  %.1f img/s %.2f ms/step' % (d['value'], d['ms_per_step']))"; }
for b in 16 32 64 128; do
for r in 1 2; do
  echo "B=$b default (one-round rule)"; EXTRA="--batch $b" step PSLD_X=1
  echo "B=$b PSLD_WINOGRAD=2"; EXTRA="--batch $b" step PSLD_WINOGRAD=2
done; done
echo "CelebA-64 B=128 default"; EXTRA="--config celeba64_sota --steps 10" step PSLD_X=1
echo "CelebA-64 B=128 PSLD_WINOGRAD=2"; EXTRA="--config celeba64_sota --steps 10" step PSLD_WINOGRAD=2
