# round-5 GPU batch 4: config-5 update A/B (per-NU B image in k_wide, sticky error in k_wide_sk) against the previous library, the whole GPU suite
# with durations, then the profiling round (kernel trace, PMC passes, bench) on the final kernels
set -u
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?run through gpurun}"; O=gpurun_out/r5e; mkdir -p $O
cp distributed-drl_amd/libddrl_hip.so /tmp/new.so
for i in 1 2; do
  echo "== new"; python tools/ddqn_cfg5_prof.py 60 2>&1 | grep "ddqn update"
  cp tools/libddrl_hip_round4_wide.so distributed-drl_amd/libddrl_hip.so
  echo "== previous library"; python tools/ddqn_cfg5_prof.py 60 2>&1 | grep "ddqn update"
  cp /tmp/new.so distributed-drl_amd/libddrl_hip.so
done > $O/ddqn_ab.txt 2>&1
bash tools/ddqn_cfg5_prof.sh r5e/cfg5 > $O/cfg5_prof.log 2>&1
python -m pytest tests -q -m gpu --durations=15 --deselect tests/test_gpu_bench_line.py::test_single_gpu_line > $O/t_all.log 2>&1; echo rc=$? >> $O/t_all.log
bash tools/prof_round.sh r5e/prof > $O/prof_round.log 2>&1
cat $O/ddqn_ab.txt; tail -n 25 $O/t_all.log; head -n 12 $O/prof/kernel_trace_summary.txt; head -c 600 $O/prof/bench.json
