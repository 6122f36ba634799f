set -u
cd "${GRAFT_REPO_ROOT:?}"; O=gpurun_out/r6_b4; mkdir -p $O
python scratch/overlap_probe.py 2>&1 | grep -v amdgpu.ids
UPG=0 python scratch/overlap_probe.py 2>&1 | grep -v amdgpu.ids | head -2
GPU_MAX_HW_QUEUES=8 python scratch/overlap_probe.py 2>&1 | grep -v amdgpu.ids | head -2
