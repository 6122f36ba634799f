# L2 (TCC) hit / miss / fabric-read counters per launch of the update, in situ (eager updates back to back).
set -eu
grep() { command grep "$@" || true; }   # display filters: no match is not an error under set -e
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}"; O=gpurun_out/${1:-r3pmc}; mkdir -p $O
for c in "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_EA0_RDREQ_sum" "TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum"; do
  tag=$(echo $c | tr ' ' '_')
  rocprofv3 --pmc $c --output-format csv -d $O/pmc_$tag -- python3 tools/insitu.py 40 > $O/pmc_$tag.log 2>&1
  python3 tools/pmc_summary.py $O/pmc_$tag > $O/pmc_$tag.txt 2>&1; rm -rf $O/pmc_$tag
  grep "k_d" $O/pmc_$tag.txt
done
