# The round-end check on the GPU box (gpurun): the GPU suite, smoke, determinism soaks, three fuzz seeds, the default bench.
# usage: bash tools/final_check.sh <tag>  ->  gpurun_out/<tag>/*
set -u
cd "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}"; O=gpurun_out/${1:-final}; mkdir -p $O
python -m pytest tests/ -q -m gpu > $O/gpu_tests.log 2>&1; tail -3 $O/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
( echo "== soak 200000 x 256 (50 per graph)"; python tools/soak.py 200000 256 50 2>&1 | grep -v amdgpu | tail -1
  echo "== dqn_soak 3000 ddqn"; python tools/dqn_soak.py 3000 ddqn 2>&1 | grep -v amdgpu | tail -1
  echo "== dqn_soak 2000 sqn"; python tools/dqn_soak.py 2000 sqn 2>&1 | grep -v amdgpu | tail -1 ) | tee $O/soak.txt
( for seed in 41 42 43; do echo "== DDRL_FUZZ_N=150 DDRL_FUZZ_SEED=$seed python -m pytest tests/test_gpu_fuzz_shapes.py -q -m gpu"; DDRL_FUZZ_N=150 DDRL_FUZZ_SEED=$seed python -m pytest tests/test_gpu_fuzz_shapes.py -q -m gpu 2>&1 | grep "passed\|failed\|^FAILED" | tail -3; done ) | tee $O/fuzz.txt
python bench.py > $O/bench.json 2> $O/bench.err; tail -1 $O/bench.err; head -c 300 $O/bench.json
