set -u
cd "${GRAFT_REPO_ROOT:?}"; O=gpurun_out/r6_b34; mkdir -p $O
bash tools/prof_round.sh r6p5 > $O/prof.log 2>&1; tail -1 $O/prof.log | cut -c1-200
