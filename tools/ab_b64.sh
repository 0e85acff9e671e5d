cd ${GRAFT_REPO_ROOT:-/root/repo}
step() { env "$@" python3 bench.py --steps 20 --warmup 3 --sample-batch 0 --no-cpu-baseline --no-probe --no-forward --no-config-block $EXTRA 2> /dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('  %.1f img/s %.2f ms/step' % (d['value'], d['ms_per_step']))"; }
for r in 1 2 3; do
  echo "B=64 default"; EXTRA="--batch 64" step PSLD_X=1
  echo "B=64 PSLD_WINOGRAD=2 (everything Winograd)"; EXTRA="--batch 64" step PSLD_WINOGRAD=2
  echo "B=64 PSLD_WINOGRAD=0"; EXTRA="--batch 64" step PSLD_WINOGRAD=0
done
