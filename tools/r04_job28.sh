cd ${GRAFT_REPO_ROOT:-/root/repo}
export PSLD_HIP_LIB=$PWD/psld_amd/libpsld_hip_abl.so
S="256,256,32;512,256,32;256,256,16;512,256,16"
run() { echo "== $*"; env "$@" python3 tools/bench_wino.py --rounds 5 --shapes "$S" 2>&1 | grep "conv fwd" | sed -e 's/direct fp32-in.*limb-in *[0-9.]* TF//' | cut -c1-100; }
PSLD_WINO_LA=3 python3 tools/wino_digest.py > /tmp/d3.txt; python3 tools/wino_digest.py > /tmp/d2.txt; diff /tmp/d2.txt /tmp/d3.txt && echo DIGESTS_SAME
for r in 1 2 3; do
run PSLD_WINO_LA=2
run PSLD_WINO_LA=3
done
