set -u
cd "${GRAFT_REPO_ROOT:?}"; O=gpurun_out/r6_b27; mkdir -p $O
DDRL_FUZZ_N=150 DDRL_FUZZ_SEED=31 python -m pytest tests/test_gpu_fuzz_shapes.py -q -m gpu -k "44-2-hid73-176-sqn" 2>&1 | tail -60 > $O/fail.txt
DDRL_LIB_PATH=tools/ab/libddrl_hip_r5.so DDRL_FUZZ_N=150 DDRL_FUZZ_SEED=31 python -m pytest tests/test_gpu_fuzz_shapes.py -q -m gpu -k "44-2-hid73-176-sqn" 2>&1 | tail -5 > $O/fail_r5.txt
