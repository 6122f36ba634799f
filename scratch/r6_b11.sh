set -u
cd "${GRAFT_REPO_ROOT:?}"; O=gpurun_out/r6_b11; mkdir -p $O
echo "== cfg5 stagger sweep (forward,wgrad in 64-cycle units)"
for st in "0,0" "24,0" "40,0" "64,0" "0,24" "0,40" "0,64" "40,40"; do
DDRL_WIDE_STAGGER=$st python tools/ddqn_cfg5_prof.py 60 2>&1 | grep "ddqn update" | sed "s/^/stagger $st: /"
done | tee $O/stagger.txt
