set -u
cd "${GRAFT_REPO_ROOT:?}"; O=gpurun_out/r6_b21; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -4
python bench.py > $O/bench.json 2> $O/bench.err; tail -1 $O/bench.err; python - <<'PY'
import json
d=json.load(open("gpurun_out/r6_b21/bench.json"))
print(d["value"], d["updates_per_s"], d["roofline"]["frac"], d["roofline"]["traffic_stale"], d["value_ungated"], d["updates_per_s_ungated"])
h=d["stages"]["host_surface_pcie_inclusive"]; print(h["sample_plus_train_per_s"], h["sample_plus_train_prefetch_per_s"])
print(d["stages"]["ddqn_update_cfg5"]["ms"])
PY
