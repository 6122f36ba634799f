set -u
cd "${GRAFT_REPO_ROOT:?}"; O=gpurun_out/r6_b12; mkdir -p $O
timeout 300 tools/wide_bench.bin 2>&1 | tail -14 | tee $O/wide_bench.txt
timeout 300 tools/wide_bench_st.bin 2>&1 | tail -8 | tee $O/wide_bench_st.txt
