set -u
cd "${GRAFT_REPO_ROOT:?}"; O=gpurun_out/r6_b16; mkdir -p $O
python -m pytest tests/test_partition.py -q -m gpu -k "free_running" 2>&1 | tail -4
python bench.py --gpus 2 --gate free --steps 1 --warmup 1 --gpu-seconds 0 --no-stages --no-cpu-baseline > $O/bench2_free.json 2> $O/bench2_free.err; tail -2 $O/bench2_free.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r6_b16/bench2_free.json"))
print({k:d[k] for k in ("n_gpus","functional_only","value","ms_per_step")}, d["functional_value"], d["config"]["gate"], d["config"]["updates_per_step"], d["other_gate"])
print(d["partition_stats"])
PY
python bench.py --gpus 2 --steps 1 --warmup 1 --gpu-seconds 0 --no-stages --no-cpu-baseline > $O/bench2_hold.json 2> $O/bench2_hold.err; tail -2 $O/bench2_hold.err
python -m pytest tests/test_gpu_bench_line.py -q -m gpu 2>&1 | tail -3
