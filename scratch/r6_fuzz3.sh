set -u
cd "${GRAFT_REPO_ROOT:?}"
python scratch/fuzz_diag.py 2560 3 68 340 195 2>&1 | grep -v amdgpu
DDRL_WIDE_SK=0 python scratch/fuzz_diag.py 2560 3 68 340 195 2>&1 | grep -v amdgpu
python scratch/fuzz_diag.py 2398 6 280 240 192 2>&1 | grep -v amdgpu
python scratch/fuzz_diag.py 2400 6 280 240 192 2>&1 | grep -v amdgpu
