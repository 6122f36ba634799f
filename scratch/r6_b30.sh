set -u
cd "${GRAFT_REPO_ROOT:?}"
python scratch/hs_split.py 2>&1 | grep -v amdgpu.ids | head -4
timeout 900 python -m pytest tests/test_gpu_replay.py tests/test_gpu_driver.py -q -m gpu -x -k "prefetch or cache or Cache or host" 2>&1 | tail -4
python tools/host_surface.py 2>&1 | grep -v amdgpu.ids | tail -12
