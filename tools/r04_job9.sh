cd ${GRAFT_REPO_ROOT:-/root/repo}
export PSLD_HIP_LIB=$PWD/psld_amd/libpsld_hip_abl.so PSLD_WINO_ABL=64
python3 tools/wino_stamps.py 256 256 32 128
python3 tools/wino_stamps.py 512 256 32 128
