# round-5 GPU batch: does the benched configuration train?  
set -u
cd "${GRAFT_REPO_ROOT:?run through gpurun}"; O=gpurun_out/r5learn; mkdir -p $O
run() { echo "== python tools/learn_check.py $*"; python tools/learn_check.py "$@" 2>&1 | grep -v amdgpu.ids; echo; }
{
run --preset sac1 --envs 4096 --seconds 360
run --preset lander --envs 4096 --seconds 360
run --preset dsac --envs 4096 --seconds 360
} > $O/curves_4096.txt
{
run --preset sac1 --envs 256 --seconds 120 --windows 12
run --preset lander --envs 256 --seconds 120 --windows 12
run --preset sac1 --envs 256 --seconds 120 --windows 12 --gamma 0.99
run --preset sac1 --envs 256 --seconds 120 --windows 12 --alpha 0.2
run --preset sac1 --envs 256 --seconds 120 --windows 12 --alpha 0.2 --gamma 0.99
run --preset lander --envs 1024 --seconds 180 --windows 12
} > $O/curves_small.txt
grep -E "==|best" $O/curves_4096.txt $O/curves_small.txt
