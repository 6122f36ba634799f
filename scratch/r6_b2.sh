set -u
cd "${GRAFT_REPO_ROOT:?}"; O=gpurun_out/r6_b2; mkdir -p $O
python -m pytest tests/test_gpu_rccl.py -x -q > $O/rccl.log 2>&1; tail -3 $O/rccl.log
python scratch/capture_control.py > $O/capture_control.log 2>&1; grep "restore OFF" $O/capture_control.log
echo "== dp step (r5 library, then this tree)"
DDRL_LIB_PATH=$PWD/tools/ab/libddrl_hip_r5.so python tools/dp_host_cost.py 2>&1 | grep "us/update" | sed 's/^/r5  /'
python tools/dp_host_cost.py 2>&1 | grep "us/update" | sed 's/^/new /'
DDRL_LIB_PATH=$PWD/tools/ab/libddrl_hip_r5.so python tools/dp_host_cost.py 2>&1 | grep "us/update" | sed 's/^/r5  /'
python tools/dp_host_cost.py 2>&1 | grep "us/update" | sed 's/^/new /'
echo "== cfg5 A/B (r5 library vs this tree)"
for i in 1 2; do
DDRL_LIB_PATH=$PWD/tools/ab/libddrl_hip_r5.so python tools/ddqn_cfg5_prof.py 60 2>&1 | grep "ddqn update" | sed 's/^/r5  /'
python tools/ddqn_cfg5_prof.py 60 2>&1 | grep "ddqn update" | sed 's/^/new /'
done
python tools/sk_ab.py 2>&1 | grep -v amdgpu.ids | tail -6
python -m pytest tests/test_gpu_math_fixtures.py tests/test_gpu_sac1.py tests/test_gpu_fuzz_shapes.py -x -q > $O/sac_dqn_tests.log 2>&1; tail -3 $O/sac_dqn_tests.log
