cd ${GRAFT_REPO_ROOT:-/root/repo}
echo "== soak_repeat B=16 (two-stream step, Winograd-domain weight gradients by policy)"; timeout 600 python tools/soak_repeat.py --steps 100 --batch 16 2>&1 | tail -6
echo "== soak_repeat B=32"; timeout 600 python tools/soak_repeat.py --steps 60 --batch 32 2>&1 | tail -6
step() { env "$@" python3 bench.py --steps 10 --warmup 3 --sample-batch 0 --no-cpu-baseline --no-probe --no-forward --no-config-block 2> /dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('  %.1f img/s %.2f ms/step' % (d['value'], d['ms_per_step']))"; }
for r in 1 2; do echo "B=128 weight gradients on the compute stream (auto)"; step PSLD_X=1; echo "B=128 weight gradients on the side stream (PSLD_OVERLAP_WGRAD=1)"; step PSLD_OVERLAP_WGRAD=1; done
echo "== default forward block with the live HBM line"
python3 bench.py --steps 3 --warmup 2 --sample-batch 0 --no-cpu-baseline --no-probe --no-config-block 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); f=d['forward']; print(f['ms'], f['hbm_bound_aggregate_frac']); print(json.dumps(f['hbm_live'])[:1600])"
