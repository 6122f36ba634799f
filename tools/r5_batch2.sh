# round-5 GPU batch 2: tile re-split A/B on the update harness, learner / driver / env suites on the new tables and the new version plan,
# rollout with versions live (16 and 230 versions), functional 2- and 8-rank lines for the record
set -u
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?run through gpurun}"; O=gpurun_out/r5c; mkdir -p $O
for i in 1 2 3; do
  echo "== head (b8290e2 kernels)"; ./tools/upd_bench_head.bin 50
  echo "== re-split (bq 256 / mid 500 / pi 210 tiles)"; ./tools/upd_bench.bin 50
  echo "== same binary, DDRL_BQ_SPLIT=0"; DDRL_BQ_SPLIT=0 ./tools/upd_bench.bin 50
done > $O/upd_ab.txt 2>&1
./tools/upd_bench_st.bin 50 > $O/upd_anatomy_resplit.txt 2>&1
python -m pytest tests/test_gpu_sac1.py tests/test_gpu_math_fixtures.py tests/test_gpu_fuzz_shapes.py tests/test_gpu_driver.py tests/test_gpu_env.py -x -q -m gpu > $O/t_learner.log 2>&1; echo rc=$? >> $O/t_learner.log
python tools/version_step_probe.py 16 4096 > $O/vprobe16.log 2>&1
python tools/version_step_probe.py 230 4096 > $O/vprobe230.log 2>&1
python tools/version_step_probe.py 16 8192 > $O/vprobe16_8192.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/vtrace -- python3 tools/version_step_probe.py 16 4096 > $O/vprobe_traced.log 2>&1
python3 tools/trace_summary.py $O/vtrace > $O/vprobe_kernels.txt 2>&1; rm -rf $O/vtrace
python bench.py --gpus 2 --steps 1 --warmup 1 --gpu-seconds 0 --no-stages --no-cpu-baseline > $O/line_2ranks_one_gpu.json 2> $O/line_2ranks.err
python bench.py --gpus 8 --steps 1 --warmup 0 --gpu-seconds 0 --no-stages --no-cpu-baseline > $O/line_8ranks_one_gpu.json 2> $O/line_8ranks.err
grep -E "==|us/update" $O/upd_ab.txt; tail -n 3 $O/t_learner.log; grep "versions live" $O/vprobe*.log; head -n 8 $O/vprobe_kernels.txt; head -c 300 $O/line_8ranks_one_gpu.json
