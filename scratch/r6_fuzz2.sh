set -u
cd "${GRAFT_REPO_ROOT:?}"; O=gpurun_out/r6_fuzz2; mkdir -p $O
for seed in 21 23; do
echo "== r5 library, seed $seed"
DDRL_LIB_PATH=$PWD/tools/ab/libddrl_hip_r5.so DDRL_FUZZ_N=150 DDRL_FUZZ_SEED=$seed python -m pytest tests/test_gpu_fuzz_shapes.py -q -m gpu -k dqn 2>&1 | grep "passed\|failed\|^FAILED" | tail -4
done
