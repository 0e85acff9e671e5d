cd ${GRAFT_REPO_ROOT:-/root/repo}
step() { env "$@" python3 bench.py --steps 40 --warmup 6 --sample-batch 0 --no-cpu-baseline --no-probe --no-forward --no-config-block $EXTRA 2> /dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('  %.1f img/s %.2f ms/step captured=%s' % (d['value'], d['ms_per_step'], d.get('captured_step')))"; }
for b in 16 32; do for r in 1 2; do
  echo "B=$b eager"; EXTRA="--batch $b" step PSLD_X=1
  echo "B=$b hipGraph-captured step"; EXTRA="--batch $b --graphs" step PSLD_X=1
  echo "B=$b eager, no side stream"; EXTRA="--batch $b" step PSLD_OVERLAP_WGRAD=0
  echo "B=$b captured, no side stream"; EXTRA="--batch $b --graphs" step PSLD_OVERLAP_WGRAD=0
done; done
