set -u
cd "${GRAFT_REPO_ROOT:?run through gpurun}"; O=gpurun_out/r5learn2; mkdir -p $O
run() { echo "== python tools/learn_check.py $*"; python tools/learn_check.py "$@" 2>&1 | grep -v amdgpu.ids; echo; }
{
run --preset sac1 --envs 4096 --seconds 1200 --windows 40
run --preset sac1 --envs 1024 --seconds 360
run --preset sac1 --envs 4096 --seconds 360 --gamma 0.99
} > $O/curves_sac1_long.txt
grep -E "==|best" $O/curves_sac1_long.txt
