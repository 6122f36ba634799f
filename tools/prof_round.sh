# One profiling round of the update on the GPU box (gpurun): kernel trace of the bench, PMC passes over IN-SITU updates (every
# dispatch in the cache state the update sequence leaves: tools/insitu.py, eager launches — PMC collection does not survive
# hipGraph replays on this ROCm) and over the same launches repeated back to back (warm: tools/stage_times.py).
# usage: bash tools/prof_round.sh <tag>   ->  gpurun_out/<tag>/*, then `python3 tools/make_traffic.py gpurun_out/<tag> rNN` on the host
set -eu
grep() { command grep "$@" || true; }   # display filters: no match is not an error under set -e
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}"; O=gpurun_out/${1:-r3p}; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --steps 5 --warmup 2 --gpu-seconds 0 --cpu-budget 3 --no-stages > $O/bench_traced.json 2> $O/bench_traced.err
python3 tools/trace_summary.py $O/trace > $O/kernel_trace_summary.txt; cp $O/trace/*/*kernel_stats.csv $O/kernel_stats.csv; rm -rf $O/trace
for c in FETCH_SIZE WRITE_SIZE; do rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -- python3 tools/insitu.py 40 > $O/pmc_$c.log 2>&1; python3 tools/pmc_summary.py $O/pmc_$c > $O/pmc_$c.txt 2>&1; rm -rf $O/pmc_$c; done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc_sq -- python3 tools/insitu.py 40 > $O/pmc_sq.log 2>&1; python3 tools/pmc_summary.py $O/pmc_sq > $O/pmc_sq.txt 2>&1; rm -rf $O/pmc_sq
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d $O/pmc_tcc -- python3 tools/insitu.py 40 > $O/pmc_tcc.log 2>&1; python3 tools/pmc_summary.py $O/pmc_tcc > $O/pmc_tcc.txt 2>&1; rm -rf $O/pmc_tcc
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d $O/pmc_tccw -- python3 tools/stage_times.py 40 nograph > $O/pmc_tcc_warm.log 2>&1; python3 tools/pmc_summary.py $O/pmc_tccw > $O/pmc_tcc_warm.txt 2>&1; rm -rf $O/pmc_tccw
python3 bench.py > $O/bench.json 2> $O/bench.err
head -8 $O/kernel_trace_summary.txt; grep -h "k_d" $O/pmc_FETCH_SIZE.txt $O/pmc_WRITE_SIZE.txt $O/pmc_sq.txt $O/pmc_tcc.txt $O/pmc_tcc_warm.txt | head -30; head -c 400 $O/bench.json
