set -u
cd "${GRAFT_REPO_ROOT:?}"; O=gpurun_out/r6_b36; mkdir -p $O
python3 tools/dp_host_cost.py > $O/dp_host_cost.txt 2>&1; grep "us/update" $O/dp_host_cost.txt
DDRL_LIB_PATH=tools/ab/libddrl_hip_r5.so python3 tools/dp_host_cost.py 2>&1 | grep "us/update" | sed 's/^/r5  /'
timeout 1500 python -m pytest tests/test_gpu_sac1.py tests/test_gpu_rccl.py tests/test_gpu_math_fixtures.py tests/test_gpu_fuzz_shapes.py -q -m gpu -x 2>&1 | tail -4
