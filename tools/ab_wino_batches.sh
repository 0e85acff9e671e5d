#!/bin/bash
# Winograd policy check at small batches and on CelebA-64: bench lines with PSLD_WINOGRAD=1 (default policy) and 0
for b in 16 32 64; do for w in 1 0; do
  echo "B=$b PSLD_WINOGRAD=$w: $(PSLD_WINOGRAD=$w python3 bench.py --batch $b --steps 30 --warmup 5 --sample-batch 0 --no-cpu-baseline --no-probe 2>/dev/null | python3 -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"],1), "img/s", round(d["ms_per_step"],2), "ms")')"
done; done
for w in 1 0; do
  echo "celeba64 B=128 PSLD_WINOGRAD=$w: $(PSLD_WINOGRAD=$w python3 bench.py --config celeba64_sota --steps 10 --warmup 3 --sample-batch 0 --no-cpu-baseline --no-probe 2>/dev/null | python3 -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"],1), "img/s", round(d["ms_per_step"],2), "ms")')"
done
