set -u
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}"; O=gpurun_out/r6_b35; mkdir -p $O
python3 tools/dp_host_cost.py 2>&1 | grep -v amdgpu.ids | tail -4 | tee $O/dp_host_cost.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr -- python3 tools/dp_host_cost.py > $O/dp_traced.txt 2>&1
python3 tools/trace_summary.py $O/tr | head -12 | cut -c1-150 | tee $O/dp_trace_summary.txt
rm -rf $O/tr
