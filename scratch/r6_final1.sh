set -u
cd "${GRAFT_REPO_ROOT:?}"; O=gpurun_out/r6_final1; mkdir -p $O
bash tools/prof_round.sh r6p2 > $O/prof_round.log 2>&1; tail -3 $O/prof_round.log
