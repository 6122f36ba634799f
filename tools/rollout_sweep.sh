set -eu
grep() { command grep "$@" || true; }   # display filters: no match is not an error under set -e
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}"; O=gpurun_out/${1:-r3roll}; mkdir -p $O
for n in 64 1024 4096 8192 16384; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/t_$n -- python3 tools/rollout_prof.py 200 $n > $O/roll_$n.log 2>&1
  tail -1 $O/roll_$n.log; python3 tools/trace_summary.py $O/t_$n | grep "k_actor_fwd\|k_env_step_pi"; rm -rf $O/t_$n
done
