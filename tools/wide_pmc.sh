# usage (GPU box): bash tools/wide_pmc.sh <outdir>  — kernel trace + PMC passes over tools/wide_bench.bin, summaries per kernel
set -eu
grep() { command grep "$@" || true; }   # display filters: no match is not an error under set -e
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}"; O=gpurun_out/${1:?usage: pass an output tag}; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/trace -- tools/wide_bench.bin > $O/trace.log 2>&1
python3 tools/trace_summary.py $O/trace > $O/wide.txt 2>&1; rm -rf $O/trace
for c in "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVE_CYCLES" "SQ_WAVES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum"; do
  tag=$(echo $c | tr ' ' '_' | cut -c1-60)
  rocprofv3 --pmc $c --output-format csv -d $O/pmc_$tag -- tools/wide_bench.bin > $O/pmc_$tag.log 2>&1
  python3 tools/pmc_summary.py $O/pmc_$tag >> $O/wide.txt 2>&1; rm -rf $O/pmc_$tag
done
grep -v "simple_timer\|amdgpu.ids" $O/wide.txt | grep "k_wide\|counters\|k_gemm"
