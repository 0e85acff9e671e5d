cd ${GRAFT_REPO_ROOT:-/root/repo}
export PSLD_HIP_LIB=$PWD/psld_amd/libpsld_hip_abl.so
for a in 64 65 66 67; do echo "=== PSLD_WINO_ABL=$a"; PSLD_WINO_ABL=$a python3 tools/wino_stamps.py 256 256 32 128 | grep -E "waves|MFMA|transform|total"; done
