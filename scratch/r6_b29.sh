set -u
cd "${GRAFT_REPO_ROOT:?}"
python scratch/hs_split.py 2>&1 | grep -v amdgpu.ids | head -4
DDRL_HOST_GRAPH=0 python scratch/hs_split.py 2>&1 | grep -v amdgpu.ids | head -4
timeout 900 python -m pytest tests/test_gpu_sac1.py tests/test_gpu_driver.py -q -m gpu -x -k "host" 2>&1 | tail -4
