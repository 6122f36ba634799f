set -u
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}"; O=gpurun_out/r6_b24; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr -- python3 tools/version_step_probe.py 16 8192 > $O/probe.txt 2>&1
python3 tools/trace_summary.py $O/tr | head -6 | cut -c1-150
rm -rf $O/tr
rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr -- python3 tools/version_step_probe.py 16 4096 > $O/probe2.txt 2>&1
python3 tools/trace_summary.py $O/tr | head -6 | cut -c1-150
rm -rf $O/tr
