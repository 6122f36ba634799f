set -u
cd "${GRAFT_REPO_ROOT:?}"; O=gpurun_out/r6_b17; mkdir -p $O
python tools/learn_check.py --preset sac1 --free 64 --free-updates 100 --seconds 480 --windows 24 > $O/learn_sac1_free_480.txt 2>&1
tail -4 $O/learn_sac1_free_480.txt
python tools/learn_check.py --preset sac1 --free 64 --free-updates 100 --seconds 240 --windows 12 --seed 1 > $O/learn_sac1_free_240_s1.txt 2>&1
tail -2 $O/learn_sac1_free_240_s1.txt
python bench.py > $O/bench.json 2> $O/bench.err; tail -1 $O/bench.err; python - <<'PY'
import json
d=json.load(open("gpurun_out/r6_b17/bench.json"))
print(d["value"], d["updates_per_s"], d["roofline"]["frac"], d["roofline"]["traffic_stale"], d["value_ungated"], d["updates_per_s_ungated"])
h=d["stages"]["host_surface_pcie_inclusive"]; print(h["sample_plus_train_per_s"], h["sample_plus_train_prefetch_per_s"])
print(d["stages"]["ddqn_update_cfg5"]["ms"])
PY
