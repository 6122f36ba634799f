"""bench.py's contract, checked on the GPU box: the one JSON line of a short single-GPU run (metric string, roofline and config
blocks, both gate figures, a fresh traffic file) and of a functional 2-rank run on the one GPU at config 3's real sizes (4096 envs per
rank, batch 256, 500 k-transition shards) — which must say that it is not a scaling point."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
METRIC = "env-steps/s + learner updates/s, SAC1 LunarLanderContinuous-v2 @1/2/4/8 GPU"


def run_bench(*args, timeout=900):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "DDRL_DIST_FORCE", "DDRL_DIST_BACKEND"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(args), env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       timeout=timeout, cwd=ROOT)
    assert p.returncode == 0, p.stderr.decode("utf-8", "replace")[-3000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1, lines                      # ONE JSON line on stdout, nothing else
    return json.loads(lines[0])


def test_single_gpu_line():
    j = run_bench("--steps", "2", "--warmup", "1", "--gpu-seconds", "0", "--no-stages", "--no-cpu-baseline")
    assert j["metric"] == METRIC and j["unit"] == "env-steps/s" and j["n_gpus"] == 1 and j["steps"] == 2 and j["warmup"] == 1
    assert j["higher_is_better"] is True and j["scaling"] == "weak" and j["vs_baseline"] is None and j["dtype"] == "f32" and j["data"] == "synthetic"
    assert j["value"] > 1e4 and abs(j["value"] / j["updates_per_s"] - 2.0) < 1e-9       # the gate: 2 env steps per sampled batch
    assert j["value_ungated"] == j["value"] and j["functional_only"] is False           # N = 1: the same step under both gate settings
    assert abs(j["ms_per_step"] * 1e-3 * j["value"] - 4096) < 1e-6 * 4096
    r = j["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 157.3 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert 0.05 < r["frac"] < 1.0 and r["update_flops"] == 949145600
    # profiles/traffic.json was taken on THESE kernels (tools/prof_round.sh + tools/make_traffic.py after any change under csrc/)
    assert r["traffic"] and r["traffic_stale"] is False, "csrc/ changed since the PMC passes behind profiles/traffic.json: re-run tools/prof_round.sh"
    c = j["config"]
    assert "workload" in c and "model" not in c and c["weak_scaling_read_against"] == "value_ungated" and c["distinct_devices"] == 1
    assert c["rank_devices"][0]["rank"] == 0 and c["rank_devices"][0]["uuid"]


def test_two_ranks_on_one_gpu_is_functional_only():
    j = run_bench("--gpus", "2", "--steps", "1", "--warmup", "1", "--gpu-seconds", "0", "--no-stages", "--no-cpu-baseline")
    assert j["n_gpus"] == 2 and j["functional_only"] is True and j["value"] == 0.0 and "value_ungated" not in j
    f = j["functional_value"]
    assert f["value"] > 0 and f["updates_per_s"] > 0 and f["value_ungated"] > 0 and "NOT an N-GPU measurement" in f["note"]
    c = j["config"]
    assert c["backend"] == "gloo" and c["world_size"] == 2 and c["distinct_devices"] == 1 and len(c["rank_devices"]) == 2
    assert c["num_envs"] == 4096 and c["batch"] == 256 and c["learner_ranks"] == [0] and c["rollout_ranks"] == [0, 1]
    assert c["gate"] == "hold" and c["updates_per_step"] == 4096 and j["other_gate"]["gate"] == "free" and j["other_gate"]["updates_per_step"] == 2048
    ps = j["partition_stats"]                          # config 3: the learner draws about half its batches from the remote shard
    tot = ps["local_batches"] + ps["remote_batches"]
    assert tot > 0 and 0.4 < ps["remote_batches"] / tot < 0.6 and ps["pushes"] >= 1
    assert j["scaling_readout"]["roles"] == ["learner+rollout", "rollout"]
