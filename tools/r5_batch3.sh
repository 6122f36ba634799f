# round-5 GPU batch 3: after the q2 image double-buffering and the single-buffered OCC=2 actor forward
set -u
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?run through gpurun}"; O=gpurun_out/r5d; mkdir -p $O
for i in 1 2 3; do
  echo "== head (b8290e2 kernels)"; ./tools/upd_bench_head.bin 50
  echo "== re-split (bq 256 / mid 500 / pi 210 tiles)"; ./tools/upd_bench.bin 50
done > $O/upd_ab.txt 2>&1
./tools/upd_bench_st.bin 50 > $O/upd_anatomy_resplit.txt 2>&1
python -m pytest tests/test_gpu_sac1.py tests/test_gpu_math_fixtures.py tests/test_gpu_fuzz_shapes.py tests/test_gpu_driver.py tests/test_gpu_env.py tests/test_partition.py -q -m gpu > $O/t_learner.log 2>&1; echo rc=$? >> $O/t_learner.log
python tools/version_step_probe.py 16 4096 > $O/vprobe16.log 2>&1
python tools/version_step_probe.py 230 4096 > $O/vprobe230.log 2>&1
python tools/version_step_probe.py 16 8192 > $O/vprobe16_8192.log 2>&1
python tools/rollout_prof.py 200 4096 > $O/roll4096.log 2>&1
python tools/rollout_prof.py 200 8192 > $O/roll8192.log 2>&1
python tools/rollout_prof.py 200 16384 > $O/roll16384.log 2>&1
python tools/soak.py 20000 > $O/soak.log 2>&1
grep -E "==|us/update" $O/upd_ab.txt; tail -n 5 $O/t_learner.log; grep -h "versions live\|rollout-only" $O/vprobe*.log $O/roll*.log; tail -n 3 $O/soak.log
