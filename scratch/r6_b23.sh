set -u
cd "${GRAFT_REPO_ROOT:?}"; O=gpurun_out/r6_b23; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_driver.py tests/test_gpu_env.py -q -m gpu -x 2>&1 | tail -5
for a in "16 4096" "16 8192" "220 4096" "40 16384"; do timeout 300 python tools/version_step_probe.py $a 2>&1 | grep -v amdgpu.ids | tee -a $O/versions.txt; done
