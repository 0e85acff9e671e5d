cd ${GRAFT_REPO_ROOT:-/root/repo}
P=$PWD/psld_amd
run() { echo "== $*"; env "$@" python3 tools/bench_limb.py --wgrad --rounds 5 --iters 5 2>&1 | grep wgrad | grep -E "@32|@16" | cut -c1-75,105-140; }
for r in 1 2; do
run PSLD_HIP_LIB=$P/libpsld_hip_abl.so PSLD_DWGRAD_WS=0
run PSLD_HIP_LIB=$P/libpsld_hip_abl.so PSLD_DWGRAD_WS=1
run PSLD_HIP_LIB=$P/libpsld_hip_v6.so
run PSLD_HIP_LIB=$P/libpsld_hip_v12.so
done
