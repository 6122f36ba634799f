"""bench.py's contract, checked on the GPU box: the one JSON line of a short single-GPU run (metric string, roofline and config
blocks, both gate figures, the three series of the N-GPU readout, a fresh traffic file), of a functional 2-rank run on the one GPU at
config 3's real sizes (4096 envs per rank, batch 256, 500 k-transition shards) and of a functional 8-rank run at config 4's real sizes
(2 learner ranks + 6 rollout ranks x 8192 envs, 10^6 transitions over 6 shards) — which must say that they are not scaling points."""
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
    # N = 1: the ungated figures are the free-running loop's — the rollout on its own stream beside the learner's graph loop, both rates
    # from ONE timed region (workers.FreeRunningLoop; example/dsac.py:229-236 has no gate)
    fr = j["free_running"]
    assert j["functional_only"] is False and j["value_ungated"] == fr["env_steps_per_s"] and j["updates_per_s_ungated"] == fr["updates_per_s"]
    assert fr["env_steps_per_s"] > 100 * j["value"] and fr["updates_per_s"] > 0.4 * j["updates_per_s"]   # neither half starves the other
    assert fr["rollout_stream_busy"] > 0.5 and fr["learner_stream_busy"] > 0.5 and fr["steps_per_segment"] >= 4
    assert j["series"]["rollout_capacity_with_learner_env_steps_per_s"] == fr["env_steps_per_s"]
    assert abs(j["ms_per_step"] * 1e-3 * j["value"] - 4096) < 1e-6 * 4096
    r = j["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 157.3 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert 0.05 < r["frac"] < 1.0 and r["update_flops"] == 949145600
    # profiles/traffic.json was taken on THESE kernels (tools/prof_round.sh + tools/make_traffic.py after any change under csrc/)
    assert r["traffic"] and r["traffic_stale"] is False, "csrc/ changed since the PMC passes behind profiles/traffic.json: re-run tools/prof_round.sh"
    c = j["config"]
    assert "workload" in c and "model" not in c and c["weak_scaling_read_against"].startswith("series.rollout_capacity") and c["distinct_devices"] == 1
    # the three series of the N-GPU readout, at N = 1: one optimizer step per batch, the rollout rank alone far above the gated rate
    se = j["series"]
    assert se["learner_ranks"] == se["rollout_ranks"] == 1 and se["functional_only"] is False
    assert abs(j["optimizer_steps_per_s"] - j["updates_per_s"]) < 1e-9 * j["updates_per_s"]
    assert 0.9 * j["updates_per_s"] < se["learner_group_optimizer_steps_per_s"] < 1.2 * j["updates_per_s"]
    assert se["rollout_capacity_env_steps_per_s"] > 100 * j["value"] and se["gated_env_steps_per_s"] == j["value"]
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


def test_eight_ranks_on_one_gpu_config4_sizes_is_functional_only():
    """BASELINE config 4 as stated (algos/sac1/sac_ray.py:137-141,246,316-324; example/dsac.py:229-233): `bench.py --gpus 8` puts 2
    data-parallel learner ranks and 6 rollout ranks x 8192 envs with a 166 666-transition shard each on the one GPU over gloo.  Roles,
    the updates the gate owes per step under both settings, where the batches came from, pushes, and learner 0 == learner 1 bit for bit."""
    j = run_bench("--gpus", "8", "--steps", "1", "--warmup", "0", "--gpu-seconds", "0", "--no-stages", "--no-cpu-baseline", timeout=1500)
    assert j["n_gpus"] == 8 and j["functional_only"] is True and j["value"] == 0.0 and "value_ungated" not in j
    c = j["config"]
    assert c["backend"] == "gloo" and c["world_size"] == 8 and c["distinct_devices"] == 1 and len(c["rank_devices"]) == 8
    assert c["num_envs"] == 8192 and c["batch"] == 256 and c["replay_capacity"] == 10 ** 6
    assert c["learner_ranks"] == [0, 1] and c["rollout_ranks"] == [2, 3, 4, 5, 6, 7]
    assert j["scaling_readout"]["roles"] == ["learner"] * 2 + ["rollout"] * 6
    # hold: 6 x 8192 env steps / a_l_ratio 2 = 24 576 batches per step, shared by the two learners; free: config 2's 2048 per learner rank
    assert c["gate"] == "hold" and c["updates_per_step"] == 12288 and c["env_steps_per_sample"] == 2.0
    assert j["other_gate"]["gate"] == "free" and j["other_gate"]["updates_per_step"] == 2048
    assert j["other_gate"]["vector_steps_per_rollout_rank_and_step"] == 256           # the free-running mode of partition.py (free_steps)
    f = j["functional_value"]
    assert f["value"] > 0 and f["updates_per_s"] > 0 and "NOT an N-GPU measurement" in f["note"]
    # one optimizer step of the learner GROUP consumes two batches
    assert abs(j["series"]["learner_group_batches_per_s"] / j["series"]["learner_group_optimizer_steps_per_s"] - 2.0) < 1e-9
    ps = j["partition_stats"]                          # rank 0's: a dedicated learner owns no shard, every batch is remote ...
    assert ps["local_batches"] == 0 and ps["remote_batches"] >= 12288 + 2048 and ps["pushes"] >= 1 + (12288 + 2048) // 300
    by = {int(k): v for k, v in ps["remote_by_owner"].items()}
    assert sorted(by) == [2, 3, 4, 5, 6, 7] and sum(by.values()) == ps["remote_batches"]
    for o, n in by.items():                            # ... one sixth from each owner (np.random.choice(6) per batch)
        assert abs(n / ps["remote_batches"] - 1 / 6) < 0.02, by
    assert ps["learners_identical"] is True and len(ps["learner_weight_crc32"]) == 2
    cap = j["series"]["rollout_capacity"]["per_rank_env_steps_per_s"]
    assert cap[:2] == [None, None] and all(v and v > 0 for v in cap[2:])
