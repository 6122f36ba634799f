# round-5 GPU batch 5: robustness record on the final kernels — shape fuzz, determinism soaks, race hunts under load
set -u
cd "${GRAFT_REPO_ROOT:?run through gpurun}"; O=gpurun_out/r5f; mkdir -p $O
{
for seed in 5 7 9; do
  echo "== DDRL_FUZZ_N=150 DDRL_FUZZ_SEED=$seed python -m pytest tests/test_gpu_fuzz_shapes.py -q -m gpu"
  DDRL_FUZZ_N=150 DDRL_FUZZ_SEED=$seed timeout 900 python -m pytest tests/test_gpu_fuzz_shapes.py -q -m gpu 2>&1 | tail -n 3
done
} > $O/fuzz.txt 2>&1
{
echo "== soak 200000 x 256 (50 per graph)"; timeout 600 python tools/soak.py 200000 256 50 2>&1 | tail -n 1
echo "== soak 20000 x 100 (16 per graph)"; timeout 300 python tools/soak.py 20000 100 16 2>&1 | tail -n 1
echo "== dqn_soak 5000 ddqn"; timeout 600 python tools/dqn_soak.py 5000 ddqn 2>&1 | tail -n 1
echo "== dqn_soak 3000 sqn"; timeout 600 python tools/dqn_soak.py 3000 sqn 2>&1 | tail -n 1
echo "== race_hunt 300 updates, batch 256, graph 10, under load"; timeout 600 python tools/race_hunt.py 300 256 10 1 2>&1 | tail -n 2
echo "== race_hunt 200 updates, batch 256, eager, under load"; timeout 600 python tools/race_hunt.py 200 256 0 1 2>&1 | tail -n 2
echo "== part_det b (feed plan over a second ring's blocks), 6 reps, graph 4"; timeout 600 python tools/part_det.py b 6 4 2>&1 | tail -n 2
} > $O/soak.txt 2>&1
cat $O/fuzz.txt $O/soak.txt
