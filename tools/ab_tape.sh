#!/bin/bash
# Training step at small per-GPU batches: eager (+ side stream) vs captured hipGraph vs launch tape; host cost per step
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/gpurun_out/r03
cd $ROOT
for b in ${BATCHES:-16 8 4 32}; do
  for mode in "" --graphs --tape; do
    python3 bench.py --batch $b --steps 30 --warmup 5 --sample-batch 0 --no-cpu-baseline --no-probe $mode 2> /dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('B=$b mode=${mode:-eager}: %.1f img/s %.2f ms/step  tape=%s' % (d['value'], d['ms_per_step'], d.get('launch_tape')))"
  done
done
for m in "" --graphs --tape; do python3 tools/host_vs_gpu.py --batch 16 $m 2>/dev/null | tail -2; done
for g in 1 4; do echo "PSLD_SIDE_GROUP=$g"; PSLD_SIDE_GROUP=$g python3 bench.py --batch 16 --steps 30 --warmup 5 --sample-batch 0 --no-cpu-baseline --no-probe --tape 2> /dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('B=16 tape: %.1f img/s %.2f ms/step  tape=%s' % (d['value'], d['ms_per_step'], d.get('launch_tape')))"; done
