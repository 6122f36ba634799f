set -u
cd "${GRAFT_REPO_ROOT:?}"; O=gpurun_out/r6_b28; mkdir -p $O
( for seed in 31 33 34; do echo "== DDRL_FUZZ_N=150 DDRL_FUZZ_SEED=$seed python -m pytest tests/test_gpu_fuzz_shapes.py -q -m gpu"; DDRL_FUZZ_N=150 DDRL_FUZZ_SEED=$seed python -m pytest tests/test_gpu_fuzz_shapes.py -q -m gpu 2>&1 | grep "passed\|failed\|^FAILED" | tail -3; done ) | tee $O/fuzz.txt
