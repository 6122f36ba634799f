set -u
cd "${GRAFT_REPO_ROOT:?}"; O=gpurun_out/r6_learn; mkdir -p $O
echo "== free-running loop, example/dsac.py's hyper-parameters (lr 1e-3, alpha 0.2, gamma 0.99), 4096 envs, 64 vector steps beside 100 updates per segment, 150 s" | tee $O/learn_free.txt
python tools/learn_check.py --preset dsac --free 64 --free-updates 100 --seconds 150 --windows 15 2>&1 | grep -v amdgpu | tee -a $O/learn_free.txt
echo "== free-running loop, the same at 256 envs, 16 vector steps beside 16 updates per segment, 90 s" | tee -a $O/learn_free.txt
python tools/learn_check.py --preset dsac --envs 256 --free 16 --free-updates 16 --seconds 90 --windows 9 2>&1 | grep -v amdgpu | tee -a $O/learn_free.txt
