set -u
cd "${GRAFT_REPO_ROOT:?}"; O=gpurun_out/r6_b9; mkdir -p $O
python -m pytest tests/test_gpu_sac1.py -x -q -k "host_batch" > $O/host.log 2>&1; tail -3 $O/host.log
python tools/host_surface.py 2>&1 | grep -v amdgpu.ids | tee $O/host_surface.txt
DDRL_HOST_GRAPH=0 python tools/host_surface.py 2>&1 | grep -v amdgpu.ids | head -3 | sed 's/^/eager: /' | tee -a $O/host_surface.txt
python -m pytest tests/test_gpu_bench_line.py -x -q > $O/benchline.log 2>&1; tail -5 $O/benchline.log
