# usage: bash tools/pmc_any.sh <outdir> "<counters...>"  — one rocprofv3 --pmc pass over tools/insitu.py, summary per kernel
set -eu
grep() { command grep "$@" || true; }   # display filters: no match is not an error under set -e
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}"; O=gpurun_out/${1:?usage: pass an output tag}; mkdir -p $O; shift
for c in "$@"; do
  tag=$(echo $c | tr ' ' '_' | cut -c1-60)
  rocprofv3 --pmc $c --output-format csv -d $O/pmc_$tag -- python3 tools/insitu.py 40 > $O/pmc_$tag.log 2>&1
  python3 tools/pmc_summary.py $O/pmc_$tag > $O/pmc_$tag.txt 2>&1; rm -rf $O/pmc_$tag
  grep "k_d\|counters" $O/pmc_$tag.txt
done
