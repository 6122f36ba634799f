set -u
cd "${GRAFT_REPO_ROOT:?}"; O=gpurun_out/r6_b20; mkdir -p $O
bash tools/prof_round.sh r6p3 > $O/prof.log 2>&1; tail -3 $O/prof.log | cut -c1-300
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -6
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
