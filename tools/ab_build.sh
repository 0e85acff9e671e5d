#!/bin/bash
# same-box A/B of two builds of the library: psld_amd/libpsld_hip_prev.so (PSLD_HIP_LIB) vs the in-tree one
cd ${GRAFT_REPO_ROOT:-/root/repo}
PREV=$PWD/psld_amd/libpsld_hip_prev.so
run() {
  env "$@" python3 bench.py --steps 10 --warmup 3 --sample-steps 30 --no-cpu-baseline --no-probe 2> /dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); s=d.get('sampling') or {}
print('  %.1f img/s %.2f ms/step   sampling batch (scaled) %.2f s' % (d['value'], d['ms_per_step'], s.get('measured_batch_s', 0) * 1000.0 / max(1, s.get('n_discrete_steps', 1000))))"
}
for r in 1 2 3; do
  echo "previous build"; run PSLD_HIP_LIB=$PREV
  echo "current build"; run PSLD_X=1
done
for r in 1 2; do
  echo "previous"; PSLD_HIP_LIB=$PREV python3 tools/bench_wino.py --rounds 3 --shapes "256,256,32;512,256,32;256,256,16" 2>&1 | grep "conv fwd" | cut -c1-150
  echo "current"; python3 tools/bench_wino.py --rounds 3 --shapes "256,256,32;512,256,32;256,256,16" 2>&1 | grep "conv fwd" | cut -c1-150
done
