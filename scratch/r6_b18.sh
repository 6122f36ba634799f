set -u
cd "${GRAFT_REPO_ROOT:?}"; O=gpurun_out/r6_b18; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_driver.py -q -m gpu -x -k "versioned or version" 2>&1 | tail -8
for a in "16 4096" "16 8192" "220 4096" "4 8192" "40 16384"; do timeout 300 python tools/version_step_probe.py $a 2>&1 | grep -v amdgpu.ids | tee -a $O/versions.txt; done
DDRL_LIB_PATH=tools/ab/libddrl_hip_r5.so python tools/version_step_probe.py 16 8192 2>&1 | grep -v amdgpu.ids | sed 's/^/r5  /' | tee -a $O/versions.txt
DDRL_LIB_PATH=tools/ab/libddrl_hip_r5.so python tools/version_step_probe.py 16 4096 2>&1 | grep -v amdgpu.ids | sed 's/^/r5  /' | tee -a $O/versions.txt
