set -u
cd "${GRAFT_REPO_ROOT:?}"; O=gpurun_out/r6_b7; mkdir -p $O
python -m pytest tests/test_gpu_sac1.py tests/test_gpu_replay.py tests/test_gpu_driver.py -x -q -k "host_batch or prefetch or cache or poison or stream_k" > $O/tests.log 2>&1; tail -4 $O/tests.log
python tools/host_surface.py 2>&1 | grep -v amdgpu.ids | tee $O/host_surface.txt
