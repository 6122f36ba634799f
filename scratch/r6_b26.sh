set -u
cd "${GRAFT_REPO_ROOT:?}"; O=gpurun_out/r6_b26; mkdir -p $O
( for seed in 31 32; do echo "== DDRL_FUZZ_N=150 DDRL_FUZZ_SEED=$seed python -m pytest tests/test_gpu_fuzz_shapes.py -q -m gpu"; DDRL_FUZZ_N=150 DDRL_FUZZ_SEED=$seed python -m pytest tests/test_gpu_fuzz_shapes.py -q -m gpu 2>&1 | grep "passed\|failed\|^FAILED" | tail -3; done ) | tee $O/fuzz.txt
python tools/learn_check.py --preset dsac --free 64 --free-updates 100 --seconds 30 --windows 6 2>&1 | grep -v amdgpu.ids > $O/learn_free_dsac.txt; tail -3 $O/learn_free_dsac.txt
python tools/learn_check.py --preset dsac --seconds 100 --windows 5 2>&1 | grep -v amdgpu.ids > $O/learn_gated_dsac.txt; tail -3 $O/learn_gated_dsac.txt
python tools/learn_check.py --preset dsac --envs 8192 --free 64 --free-updates 100 --seconds 30 --windows 6 2>&1 | grep -v amdgpu.ids > $O/learn_free_dsac_8192.txt; tail -3 $O/learn_free_dsac_8192.txt
