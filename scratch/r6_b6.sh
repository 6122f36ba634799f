set -u
cd "${GRAFT_REPO_ROOT:?}"; O=gpurun_out/r6_b6; mkdir -p $O
python -m pytest tests/test_gpu_sac1.py tests/test_gpu_driver.py tests/test_gpu_math_fixtures.py -x -q > $O/tests.log 2>&1; tail -4 $O/tests.log
python tools/host_surface.py 2>&1 | grep -v amdgpu.ids | tee $O/host_surface.txt
python bench.py --steps 5 --warmup 2 --gpu-seconds 0 --cpu-budget 2 --no-stages > $O/bench_quick.json 2> $O/bench_quick.err; tail -3 $O/bench_quick.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r6_b6/bench_quick.json"))
print("value", d["value"], "updates/s", d["updates_per_s"], "ungated", d.get("value_ungated"), d.get("updates_per_s_ungated"))
fr=d.get("free_running"); fr.pop("what",None); print(json.dumps(fr, indent=1)[:2500])
PY
