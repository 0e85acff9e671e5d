cd ${GRAFT_REPO_ROOT:-/root/repo}
ABL=$PWD/psld_amd/libpsld_hip_abl.so
PSLD_HIP_LIB=$ABL python3 tools/wino_digest.py > /tmp/d0.txt; PSLD_HIP_LIB=$ABL PSLD_WINO_ERAW=1 python3 tools/wino_digest.py > /tmp/d1.txt; diff /tmp/d0.txt /tmp/d1.txt && echo DIGESTS_SAME
S="256,256,32;512,256,32;256,256,16;512,256,16;256,256,8"
run() { echo "== $*"; env "$@" python3 tools/bench_wino.py --rounds 5 --shapes "$S" $EXTRA 2>&1 | grep "conv fwd" | sed -e 's/direct fp32-in.*limb-in *[0-9.]* TF//' | cut -c1-110; }
for r in 1 2 3; do
run PSLD_HIP_LIB=$ABL PSLD_WINO_ERAW=0
run PSLD_HIP_LIB=$ABL PSLD_WINO_ERAW=1
done
EXTRA="--batch 512"
run PSLD_HIP_LIB=$ABL PSLD_WINO_ERAW=0
run PSLD_HIP_LIB=$ABL PSLD_WINO_ERAW=1
