set -u
cd "${GRAFT_REPO_ROOT:?}"; O=gpurun_out/r6_learn; mkdir -p $O
echo "== free-running loop, example/dsac.py's hyper-parameters, 4096 envs, 64 vector steps beside 100 updates per segment, 30 s in 2-s windows" | tee $O/learn_free2.txt
python tools/learn_check.py --preset dsac --free 64 --free-updates 100 --seconds 30 --windows 15 2>&1 | grep -v amdgpu | tee -a $O/learn_free2.txt
echo "== free-running loop, algos/sac1/hyperparams.py's values (lr 5e-5, alpha 0.1, gamma 0.997: what bench.py carries), 4096 envs, 120 s" | tee -a $O/learn_free2.txt
python tools/learn_check.py --preset sac1 --free 64 --free-updates 100 --seconds 120 --windows 12 2>&1 | grep -v amdgpu | tee -a $O/learn_free2.txt
