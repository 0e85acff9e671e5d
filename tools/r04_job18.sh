cd ${GRAFT_REPO_ROOT:-/root/repo}
python3 -m pytest tests/test_kernels_gpu.py -q -x -k "groupnorm or gn_" 2>&1 | grep -E "passed|failed" | tail -1
cp psld_amd/libpsld_hip_abl.so /tmp/libpsld_hip_prev.so
for r in 1 2 3; do
echo "previous gn_bwd"; PSLD_HIP_LIB=/tmp/libpsld_hip_prev.so python3 tools/bench_hbm.py 2>&1 | grep -i -E "bwd|backward"
echo "two-stage reduction"; python3 tools/bench_hbm.py 2>&1 | grep -i -E "bwd|backward"
done
run() { env "$@" python3 bench.py --steps 10 --warmup 3 --sample-batch 0 --no-cpu-baseline --no-probe 2> /dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('  %.1f img/s %.2f ms/step' % (d['value'], d['ms_per_step']))"; }
for r in 1 2; do
  echo "previous build"; run PSLD_HIP_LIB=/tmp/libpsld_hip_prev.so
  echo "this build"; run PSLD_X=1
done
