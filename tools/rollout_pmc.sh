set -eu
grep() { command grep "$@" || true; }   # display filters: no match is not an error under set -e
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}"; O=gpurun_out/${1:-r3roll}; mkdir -p $O
for c in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $c | tr ' ' '_' | cut -c1-50)
  rocprofv3 --pmc $c --output-format csv -d $O/p_$tag -- python3 tools/rollout_prof.py 100 4096 > $O/p_$tag.log 2>&1
  python3 tools/pmc_summary.py $O/p_$tag > $O/p_$tag.txt 2>&1; rm -rf $O/p_$tag
  grep "k_actor_f\|k_env_ste\|k_version\|counters" $O/p_$tag.txt
done
