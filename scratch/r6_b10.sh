set -u
cd "${GRAFT_REPO_ROOT:?}"; O=gpurun_out/r6_b10; mkdir -p $O
python -m pytest tests/test_gpu_replay.py tests/test_gpu_driver.py tests/test_gpu_env.py -x -q > $O/tests.log 2>&1; tail -4 $O/tests.log
echo "== versioned vector step (r5 library, then this tree)"
for cfg in "16 4096" "16 8192" "224 4096"; do
DDRL_LIB_PATH=$PWD/tools/ab/libddrl_hip_r5.so python tools/version_step_probe.py $cfg 2>&1 | grep "versions live" | sed 's/^/r5  /'
python tools/version_step_probe.py $cfg 2>&1 | grep "versions live" | sed 's/^/new /'
done | tee $O/versions.txt
python tools/host_surface.py 2>&1 | grep -v amdgpu.ids | tee $O/host_surface.txt
