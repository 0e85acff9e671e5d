#!/bin/bash
# pointwise limb GEMMs: previous build vs straight-line epilogue (four-wave kernel) vs eight-wave persistent kernel
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
run() {
  env "$@" python3 bench.py --steps 10 --warmup 3 --sample-steps 30 --no-cpu-baseline --no-probe 2> /dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); s=d.get('sampling') or {}
print('  %.1f img/s %.2f ms/step   sampling batch (scaled) %s s' % (d['value'], d['ms_per_step'], '%.2f' % (s.get('measured_batch_s', 0) * 1000.0 / max(1, s.get('n_discrete_steps', 1000)))))"
}
for r in 1 2; do
  echo "previous build"; run PSLD_HIP_LIB=$ROOT/psld_amd/libpsld_hip_prev.so
  echo "new epilogue, PSLD_PW8=0"; run PSLD_PW8=0
  echo "new epilogue + eight-wave pointwise"; run PSLD_PW8=1
done
