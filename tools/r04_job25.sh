cd ${GRAFT_REPO_ROOT:-/root/repo}
export PSLD_HIP_LIB=$PWD/psld_amd/libpsld_hip_abl.so
S="256,256,32;512,256,32;256,256,16"
run() { echo "== $*"; env "$@" python3 tools/bench_wino.py --rounds 5 --shapes "$S" 2>&1 | grep "conv fwd" | sed -e 's/direct fp32-in.*limb-in *[0-9.]* TF//' | cut -c1-100; }
for r in 1 2; do
run PSLD_WINO_ABL=0
run PSLD_WINO_ABL=256
run PSLD_WINO_ABL=4
done
