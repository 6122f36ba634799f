# usage (GPU box): bash tools/ddqn_cfg5_prof.sh <outdir>  — kernel trace + PMC passes over config 5's Double-DQN update (tools/ddqn_cfg5_prof.py)
set -eu
grep() { command grep "$@" || true; }   # display filters: no match is not an error under set -e
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}"; O=gpurun_out/${1:?usage: pass an output tag}; mkdir -p $O
echo "config 5's learner (Double-DQN update, obs 28 224, batch 512, hidden (400, 300)), tools/ddqn_cfg5_prof.py 30, eager:" > $O/ddqn_cfg5.txt
python3 tools/ddqn_cfg5_prof.py 50 2>&1 | grep "ddqn update" >> $O/ddqn_cfg5.txt
rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 tools/ddqn_cfg5_prof.py 30 > $O/trace.log 2>&1
python3 tools/trace_summary.py $O/trace >> $O/ddqn_cfg5.txt 2>&1; rm -rf $O/trace
for c in "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $c | tr ' ' '_' | cut -c1-60)
  rocprofv3 --pmc $c --output-format csv -d $O/pmc_$tag -- python3 tools/ddqn_cfg5_prof.py 12 > $O/pmc_$tag.log 2>&1
  python3 tools/pmc_summary.py $O/pmc_$tag >> $O/ddqn_cfg5.txt 2>&1; rm -rf $O/pmc_$tag
done
grep -v "simple_timer\|amdgpu.ids" $O/ddqn_cfg5.txt | head -60
