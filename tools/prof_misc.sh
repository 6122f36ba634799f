# The round's micro-benchmark / side measurements, as text files under gpurun_out/<tag>/ (copy what is cited into profiles/).
set -eu
grep() { command grep "$@" || true; }   # display filters: no match is not an error under set -e
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}"; O=gpurun_out/${1:-r3m}; mkdir -p $O
tools/l2_persist_bench.bin > $O/l2_persist.txt 2>&1
hipcc --offload-arch=gfx950 -O2 -Wno-unused-result -o /tmp/xcc tools/xcc_bench.hip 2>/dev/null && /tmp/xcc > $O/xcc_placement.txt 2>&1
python3 tools/dp_host_cost.py 2>&1 | grep "us/update" > $O/dp_host_cost.txt
tools/upd_bench.bin 50 > $O/upd_anatomy.txt 2>&1; tools/upd_bench_st.bin 50 >> $O/upd_anatomy.txt 2>&1
{ echo "rollout-only vector steps (policy forward + env.step + store), rocprofv3 --kernel-trace, by number of envs:";
  for n in 64 1024 4096 8192 16384; do
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/t_$n -- python3 tools/rollout_prof.py 200 $n > $O/roll_$n.log 2>&1
    tail -1 $O/roll_$n.log; python3 tools/trace_summary.py $O/t_$n | grep "k_actor_fwd\|k_env_step_pi"; rm -rf $O/t_$n $O/roll_$n.log
  done; } > $O/rollout_sweep.txt 2>&1
{ echo "PMC passes over rollout-only steps at 4096 envs (tools/rollout_prof.py 100 4096), mean per dispatch:";
  for c in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    rocprofv3 --pmc $c --output-format csv -d $O/p -- python3 tools/rollout_prof.py 100 4096 > $O/p.log 2>&1
    python3 tools/pmc_summary.py $O/p 2>&1 | grep "k_actor_f\|k_env_ste"; rm -rf $O/p $O/p.log
  done; } > $O/rollout_pmc.txt 2>&1
cat $O/dp_host_cost.txt; tail -3 $O/rollout_sweep.txt; head -3 $O/l2_persist.txt
