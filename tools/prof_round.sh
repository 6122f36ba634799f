cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r2p; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --steps 5 --warmup 2 --gpu-seconds 0 --cpu-budget 3 > $O/bench_traced.json 2> $O/bench_traced.err
python3 tools/trace_summary.py $O/trace > $O/kernel_trace_summary.txt; cp $O/trace/*/*kernel_stats.csv $O/kernel_stats.csv; rm -rf $O/trace
for c in FETCH_SIZE WRITE_SIZE; do rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -- python3 tools/stage_times.py 20 nograph > $O/pmc_$c.log 2>&1; python3 tools/pmc_summary.py $O/pmc_$c > $O/pmc_$c.txt 2>&1; rm -rf $O/pmc_$c; done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc_sq -- python3 tools/stage_times.py 20 nograph > $O/pmc_sq.log 2>&1; python3 tools/pmc_summary.py $O/pmc_sq > $O/pmc_sq.txt 2>&1; rm -rf $O/pmc_sq
python3 bench.py > $O/bench.json 2> $O/bench.err
head -8 $O/kernel_trace_summary.txt; grep -h "k_d" $O/pmc_FETCH_SIZE.txt $O/pmc_WRITE_SIZE.txt $O/pmc_sq.txt | head -20; head -c 400 $O/bench.json
