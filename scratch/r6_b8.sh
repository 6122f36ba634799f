set -u
cd "${GRAFT_REPO_ROOT:?}"; O=gpurun_out/r6_b8; mkdir -p $O
python -m pytest tests/test_partition.py -x -q > $O/partition.log 2>&1; tail -5 $O/partition.log
python -m pytest tests/test_gpu_bench_line.py -x -q > $O/benchline.log 2>&1; tail -5 $O/benchline.log
