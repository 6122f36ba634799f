set -u
cd "${GRAFT_REPO_ROOT:?}"; O=gpurun_out/r6_b5; mkdir -p $O
python scratch/free_timeline.py 2>&1 | grep -v amdgpu.ids
K=60 python scratch/free_timeline.py 2>&1 | grep -v amdgpu.ids
python -m pytest tests/test_gpu_driver.py -x -q > $O/driver.log 2>&1; tail -4 $O/driver.log
python tools/host_surface.py 2>&1 | grep -v amdgpu.ids | tee $O/host_surface.txt
