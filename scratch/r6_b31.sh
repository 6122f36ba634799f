set -u
cd "${GRAFT_REPO_ROOT:?}"
timeout 1200 python -m pytest tests/test_gpu_replay.py tests/test_gpu_driver.py tests/test_gpu_integration_doc.py -q -m gpu -x 2>&1 | tail -4
python tools/host_surface.py 2>&1 | grep -v amdgpu.ids | head -3
