set -u
cd "${GRAFT_REPO_ROOT:?}"; O=gpurun_out/r6_b15; mkdir -p $O
python bench.py --gpus 2 --gate free --steps 1 --warmup 1 --gpu-seconds 0 --no-stages --no-cpu-baseline > $O/bench2_free.json 2> $O/bench2_free.err; tail -2 $O/bench2_free.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r6_b15/bench2_free.json"))
print({k:d[k] for k in ("n_gpus","functional_only","value","ms_per_step")}, d["functional_value"], d["config"]["gate"], d["config"]["updates_per_step"], d["other_gate"])
print(d["partition_stats"])
PY
python bench.py > $O/bench.json 2> $O/bench.err; tail -1 $O/bench.err; python - <<'PY'
import json
d=json.load(open("gpurun_out/r6_b15/bench.json"))
print(d["value"], d["updates_per_s"], d["roofline"]["frac"], d["roofline"]["traffic_stale"], d["value_ungated"], d["updates_per_s_ungated"])
print(d["stages"]["host_surface_pcie_inclusive"])
print(d["stages"]["ddqn_update_cfg5"]["ms"])
PY
