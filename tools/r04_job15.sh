cd ${GRAFT_REPO_ROOT:-/root/repo}
python3 -m pytest tests/test_kernels_gpu.py -q -x -k "wgrad or weight_grad" 2>&1 | tail -2
python3 -m pytest tests/test_model_gpu.py tests/test_blocks_gpu.py -q -x 2>&1 | tail -2
ABL=$PWD/psld_amd/libpsld_hip_abl.so
run() { env "$@" python3 bench.py --steps 10 --warmup 3 --sample-batch 0 --no-cpu-baseline --no-probe 2> /dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('  %.1f img/s %.2f ms/step' % (d['value'], d['ms_per_step']))"; }
# bench.py refuses the ablation library by name: copy it under a neutral name for this A/B of two product-equivalent kernels
cp $ABL /tmp/libpsld_hip_prev.so
for r in 1 2 3; do
  echo "round-3 dwgrad (PSLD_DWGRAD_WS=0)"; run PSLD_HIP_LIB=/tmp/libpsld_hip_prev.so PSLD_DWGRAD_WS=0
  echo "wave-specialised dwgrad"; run PSLD_X=1
done
