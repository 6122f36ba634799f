# round-5 GPU batch 1: update-harness A/B (HEAD kernels vs refactored, deferral off / on), tile-count sweep, deferral parity tests, rollout trace
set -u
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?run through gpurun}"; O=gpurun_out/r5b; mkdir -p $O
for i in 1 2 3; do
  echo "== head"; ./tools/upd_bench_head.bin 50
  echo "== new, deferral off"; ./tools/upd_bench.bin 50
  echo "== new, deferral on"; DDRL_DEFER_QW=1 ./tools/upd_bench.bin 50
done > $O/upd_ab.txt 2>&1
DDRL_DEFER_QW=1 ./tools/upd_bench_st.bin 50 > $O/upd_anatomy_defer.txt 2>&1
./tools/upd_bench_st.bin 50 > $O/upd_anatomy_plain.txt 2>&1
python -m pytest tests/test_gpu_sac1.py -x -q -m gpu -k "graph_loop or two_thousand" > $O/t_sac1.log 2>&1; echo rc=$? >> $O/t_sac1.log
python tools/tile_sweep.py 200 > $O/tile_sweep.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/vtrace -- python3 tools/version_step_probe.py 16 4096 > $O/vprobe.log 2>&1
python3 tools/trace_summary.py $O/vtrace > $O/vprobe_kernels.txt 2>&1; rm -rf $O/vtrace
cat $O/upd_ab.txt | grep -E "==|us/update"; tail -n 3 $O/t_sac1.log; tail -n 20 $O/tile_sweep.txt; head -n 20 $O/vprobe_kernels.txt; tail -n 2 $O/vprobe.log
