set -u
cd "${GRAFT_REPO_ROOT:?}"; O=gpurun_out/r6_b13; mkdir -p $O
echo "== loader wave ON"; timeout 300 tools/wide_bench.bin 2>&1 | tail -16 | tee $O/wide_bench_loader.txt
echo "== loader wave OFF"; DDRL_WIDE_LOADER=0 timeout 300 tools/wide_bench.bin 2>&1 | tail -10 | tee $O/wide_bench_noloader.txt
for i in 1 2; do
DDRL_WIDE_LOADER=0 python tools/ddqn_cfg5_prof.py 60 2>&1 | grep "ddqn update" | sed 's/^/loader off: /'
python tools/ddqn_cfg5_prof.py 60 2>&1 | grep "ddqn update" | sed 's/^/loader on : /'
done | tee $O/cfg5.txt
