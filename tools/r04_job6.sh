cd ${GRAFT_REPO_ROOT:-/root/repo}
ABL=$PWD/psld_amd/libpsld_hip_abl.so
python3 -m pytest tests/test_kernels_gpu.py -q -x -k "wino" 2>&1 | tail -3
PSLD_HIP_LIB=$ABL PSLD_WINO_R03=1 python3 tools/wino_digest.py > /tmp/d_r03.txt; python3 tools/wino_digest.py > /tmp/d_new.txt; diff /tmp/d_r03.txt /tmp/d_new.txt && echo DIGESTS_SAME_AS_R03
PSLD_HIP_LIB=$ABL PSLD_WINO_R03=1 python3 tools/wino_digest.py --small > /tmp/s_r03.txt; python3 tools/wino_digest.py --small > /tmp/s_new.txt; diff /tmp/s_r03.txt /tmp/s_new.txt && echo SMALL_DIGESTS_SAME_AS_R03
S="256,256,32;512,256,32;256,256,16;512,256,16;256,256,8"
run() { echo "== $*"; env "$@" python3 tools/bench_wino.py --rounds 5 --shapes "$S" $EXTRA 2>&1 | grep "conv fwd" | sed -e 's/direct fp32-in.*limb-in *[0-9.]* TF//' | cut -c1-110; }
for r in 1 2 3; do
run PSLD_HIP_LIB=$ABL PSLD_WINO_R03=1
run PSLD_X=0
done
EXTRA="--batch 512"
run PSLD_HIP_LIB=$ABL PSLD_WINO_R03=1
run PSLD_X=0
EXTRA=""
python3 tools/bench_wino.py --fused-gn --rounds 5 --batch 512 --shapes "256,256,32;512,256,32;256,256,16" 2>&1 | grep "conv fwd"
python3 tools/bench_wino.py --fused-gn --rounds 5 --batch 128 --shapes "256,256,32;256,256,16" 2>&1 | grep "conv fwd"
