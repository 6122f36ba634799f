set -u
cd "${GRAFT_REPO_ROOT:?}"; O=gpurun_out/r6_b3; mkdir -p $O
python -m pytest tests/test_gpu_driver.py -x -q -k "free_running or ratio_gate or device_style" > $O/free.log 2>&1; tail -5 $O/free.log
python tools/dp_host_cost.py 2>&1 | grep "us/update" | tee $O/dp_host_cost.txt
python bench.py --steps 5 --warmup 2 --gpu-seconds 0 --cpu-budget 2 --no-stages > $O/bench_quick.json 2> $O/bench_quick.err; tail -3 $O/bench_quick.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r6_b3/bench_quick.json"))
print("value", d["value"], "updates/s", d["updates_per_s"], "ungated", d.get("value_ungated"), d.get("updates_per_s_ungated"))
print(json.dumps(d.get("free_running"), indent=1)[:2500])
print(d["series"])
PY
