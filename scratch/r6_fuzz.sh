set -u
cd "${GRAFT_REPO_ROOT:?}"; O=gpurun_out/r6_fuzz; mkdir -p $O
for seed in 21 23; do
DDRL_FUZZ_N=150 DDRL_FUZZ_SEED=$seed python -m pytest tests/test_gpu_fuzz_shapes.py -q -m gpu 2>&1 | grep -v "^\.\|^$" | head -60 > $O/fuzz_$seed.log; head -50 $O/fuzz_$seed.log
done
