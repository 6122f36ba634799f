#!/usr/bin/env python3
"""bench.py — env-steps/s + learner updates/s of the SAC1 actor-learner hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: launched as `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`, or plainly as
     `python bench.py --gpus N`: without WORLD_SIZE in the environment this process only spawns the N ranks — before any
     GPU call — relays rank 0's JSON line and exits with the worst return code)

N = 1 — BASELINE.json configs[1]: SAC1 on the LunarLanderContinuous-v2 stand-in, 4096 vectorised envs + 1M-transition
device replay, batch 256, hidden (400, 300).  One "step" = one pass of the hot path over one batch of environments:
    rollout : batched policy forward (4096 x 8->400->300->(2,2)) -> env.step kernel -> store 4096
    learner : num_envs / a_l_ratio updates, each = MT19937 sample of 256 + gather -> SAC1 update, 50 per hipGraph
              (the reference's actor/learner gate keeps steps / sample_times <= a_l_ratio, algos/sac1/sac1.py:205;
              default a_l_ratio = 2, sac1.py:25)
    ps      : push of the flat parameter vector every 300 updates (sac1.py:149), pull by the actor
N > 1 — the north star's rank-role partitioning (distributed-drl_amd/partition.py):
    N = 2 (configs[2])  learner on rank 0, envs + a replay shard on BOTH ranks; per update the learner picks a shard on
                        its seeded stream, a remote owner gathers and sends the packed 20 KB batch point-to-point,
                        parameters travel as one RCCL broadcast every 300 updates
    N = 8 (configs[3])  ranks 0-1 learners (gradient all-reduce per update), ranks 2-7 rollout ranks x 8192 envs + shard
    other N             N // 4 learners (at least one), the rest rollout ranks
  A step = one vector env step on every rollout rank + the updates the reference's actor/learner gate owes for those env
  steps (steps / sample_times <= a_l_ratio, algos/sac1/sac1.py:25,203-207): env steps of the step / a_l_ratio batches,
  shared over the learner ranks (a data-parallel update of k learners draws k batches).  `--gate free` drops the gate
  (2048 updates per learner rank whatever the env steps: the round-2 figure, rollout scaling only).  Next to `value` the
  line carries what lets a 1 -> 8 curve be read as rollout scaling vs learner scaling: `env_steps_per_sample`, every
  rank's busy time inside the timed block, per-role phase times (one extra, drained block) and learner-only updates/s.
Synthetic data: rings pre-filled to capacity with seeded synthetic transitions; glorot / zeros weights.
Inputs are resident in HBM when timing starts.

Prints ONE JSON line on rank 0.  `value` = whole-job env-steps/s over EXACTLY --steps steps; `updates_per_s` (sampled batches consumed, all
learner ranks) and `optimizer_steps_per_s` (Adam steps of the learner GROUP: data-parallel learners make one from their batches) ride along.
`series` = the three curves an N-GPU read-out needs, measured apart: the learner group alone, every rollout rank flat out (all ranks at
once, no learner work), the gated `value`; `config.weak_scaling_why` says which the >= 0.7 target is read against.
`free_running` (N = 1): the rollout on its own stream beside the learner's graph loop on another, no gate (example/dsac.py:229-236;
workers.FreeRunningLoop): `value_ungated` / `updates_per_s_ungated` are the two rates of that ONE timed region, and
`series.rollout_capacity_with_learner_env_steps_per_s` the rollout capacity with a learner training beside it.
`roofline` prices the launches of one update (fp32 MFMA): algorithmic FLOPs of an update / launches per update, over the
average launch time measured with HIP events on the launch stream around a learner-only block of graph replays (so
launch gaps count).  `stages` adds the other rows of SURVEY §8(d): rollout-only, store, sample, the config-5 gather.
`cpu_baseline` times the oracle (CPU restatement of the reference path) on this box's host cores for a bounded sample,
BEFORE the GPU legs; the GPU legs run last and are repeated for ~10 s (`repeat_blocks`) so that a sampler sees them.
"""
import argparse
import ctypes
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_HBM_GBPS = 8000.0        # MI355X_MICROARCH.md: HBM3E spec (6 290 measured streaming)


TRAFFIC_SOURCES = ("ddrl_common.h", "gemm_core.h", "policy_row.h", "replay_device.h", "sac1_direct.h", "sac1.hip")


def kernel_source_hash():
    """sha256 over the csrc/ files the five profiled launches of the SAC1 update compile from (TRAFFIC_SOURCES, sorted by name):
    profiles/traffic.json records the one its PMC passes were taken on.  (Until round 6 the hash ran over every file of csrc/, so a
    change to the lander or to config 5's kernels marked the SAC1 update's counters stale.)"""
    import hashlib
    hsh = hashlib.sha256()
    for name in sorted(TRAFFIC_SOURCES):
        hsh.update(name.encode() + b"\0" + open(os.path.join(ROOT, "distributed-drl_amd", "csrc", name), "rb").read())
    return hsh.hexdigest()


def update_flops(B, obs, act, h1, h2):
    """1 853 800 MAC per sample -> 949 145 600 FLOP at B = 256."""
    pol = obs * h1 + h1 * h2 + 2 * h2 * act
    q = (obs + act) * h1 + h1 * h2 + h2
    fwd = 3 * pol + 5 * q
    bwd = (2 * pol - obs * h1) + q + 2 * (2 * q - (obs + act) * h1)
    return 2 * B * (fwd + bwd)


def policy_flops(obs, act, h1, h2):
    """124 400 MAC per policy-phase env step -> 248 800 FLOP."""
    return 2 * (obs * h1 + h1 * h2 + 2 * h2 * act)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--num-envs", type=int, default=None, help="envs per rollout rank (default 4096; 8192 on dedicated rollout ranks)")
    ap.add_argument("--capacity", type=int, default=10 ** 6, help="replay transitions in all (sharded over the rollout ranks)")
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--a-l-ratio", type=float, default=2.0)
    ap.add_argument("--updates-per-graph", type=int, default=50)  # divides push_freq = 300: no eager remainder between pushes
    ap.add_argument("--num-learners", type=int, default=None)
    ap.add_argument("--gate", choices=("hold", "free"), default="hold",
                    help="hold: updates per step = env steps of the step / a_l_ratio over the learner ranks (the reference's gate); "
                         "free: 2048 updates per learner rank per step whatever the env steps")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-stages", action="store_true", help="skip the rollout / store / sample / config-5 stage measurements")
    ap.add_argument("--no-free", action="store_true", help="skip the free-running (ungated, two-stream) measurement at N = 1")
    ap.add_argument("--free-segments", type=int, default=60, help="segments of the free-running measurement's timed region")
    ap.add_argument("--free-steps", type=int, default=256, help="N > 1, free-running mode: vector steps per rollout rank and step")
    ap.add_argument("--cpu-budget", type=float, default=8.0, help="seconds per CPU-baseline leg")
    ap.add_argument("--gpu-seconds", type=float, default=10.0, help="keep repeating the timed block until the GPU legs lasted this long")
    ap.add_argument("--stage-samples", type=int, default=200)
    ap.add_argument("--cfg5-capacity", type=int, default=-1,
                    help="transitions of the config-5 ring (225 804 B each); -1: as many as 90 %% of the free HBM holds (at most 1.15 M = 260 GB); 0: skip config 5")
    return ap.parse_args()


# ------------------------------------------------------------------------------------------------
# N > 1 without a launcher: spawn the ranks (nothing in this process has touched the GPU)
# ------------------------------------------------------------------------------------------------
def spawn_ranks(n):
    import torch  # device_count() does not initialise the GPU on this image
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ndev = torch.cuda.device_count()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if ndev < n:
            env.setdefault("DDRL_DIST_BACKEND", "gloo")  # several ranks per GPU: functional run only (RCCL refuses duplicate devices)
            env["DDRL_BENCH_FUNCTIONAL_ONLY"] = "1"      # the line says so and carries no scaling point
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out, _ = procs[0].communicate()
    rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    sys.stdout.write(out.decode())
    sys.stdout.flush()
    return max(abs(rc) for rc in rcs)


# ------------------------------------------------------------------------------------------------
# CPU baseline: the oracle on the host cores (kind "port")
# ------------------------------------------------------------------------------------------------
def cpu_baseline(a_l_ratio, budget_s):
    """Single process like Ray local mode (configs[0]); ring pre-filled to 10^6.  Leg 1 pins torch to one
    thread like the reference's session config (algos/sac1/actor_learner.py:110-111); leg 2 lets the learner use every core."""
    import numpy as np
    import torch
    from oracle import sac1_oracle as so
    from oracle.env_oracle import LanderOracle
    from oracle.replay_oracle import ReplayBufferOracle
    threads = torch.get_num_threads()
    cores = os.cpu_count() or 1
    cfg = so.Config()
    params = so.init_params(cfg, 0)
    rb = ReplayBufferOracle(8, 2, 10 ** 6, seed=0)
    rs = np.random.RandomState(1234)
    n0 = 10 ** 6
    rb.obs1_buf[:], rb.obs2_buf[:] = rs.randn(n0, 8), rs.randn(n0, 8)
    rb.acts_buf[:], rb.rews_buf[:] = rs.uniform(-1, 1, (n0, 2)), rs.randn(n0)
    rb.done_buf[:] = rs.rand(n0) < 0.01
    rb.ptr, rb.size, rb.steps = 0, n0, n0
    rn = np.random.RandomState(1)

    def learner_leg(nthreads):
        torch.set_num_threads(nthreads)
        learner = so.Sac1Oracle(cfg, params, torch.float32)
        t0, n_upd = time.perf_counter(), 0
        while time.perf_counter() - t0 < budget_s:
            batch = rb.sample_batch(256)
            eps = [rn.randn(256, 2).astype(np.float32) for _ in range(3)]
            learner.step(batch, *eps)
            n_upd += 1
        return (time.perf_counter() - t0) / n_upd, n_upd

    try:
        torch.set_num_threads(1)
        env = LanderOracle(1, seed=0)
        o = env.obs()
        # rollout leg: batch-of-1 policy forward + env.step + store, per transition
        t0, n_env = time.perf_counter(), 0
        while time.perf_counter() - t0 < budget_s:
            for _ in range(50):
                a = so.actor_act(cfg, params, o, rn.randn(1, 2).astype(np.float32))
                o2, r, d, o_next, _ = env.step(a)
                rb.store(o[0], a[0], r[0], o2[0], d[0])
                o = o_next
                n_env += 1
        t_env = (time.perf_counter() - t0) / n_env
        t_upd, n_upd = learner_leg(1)
        # "all cores": torch's intra-op pool on matrices this small degrades badly when oversubscribed (256 threads: one update
        # in 20 s), so the leg keeps the best of a short sweep up to every core and says which thread count that was
        budget_all, budget_s = budget_s, budget_s / 3.0
        t_upd_all, n_upd_all, nt_all = None, 0, 1
        for nt in sorted({min(cores, 8), min(cores, 32), cores}):
            tu, nu = learner_leg(nt)
            if t_upd_all is None or tu < t_upd_all:
                t_upd_all, n_upd_all, nt_all = tu, nu, nt
        budget_s = budget_all
    finally:
        torch.set_num_threads(threads)

    def combine(tu):
        return 1.0 / (t_env + tu / a_l_ratio)

    one = {"value": combine(t_upd), "unit": "env-steps/s", "cores": 1, "kind": "port",
           "updates_per_s": combine(t_upd) / a_l_ratio,
           "rollout_only_env_steps_per_s": 1.0 / t_env, "learner_only_updates_per_s": 1.0 / t_upd,
           "host_cores_available": cores,
           "sample": "%d env steps (batch-of-1 policy forward + env.step + store) and %d updates (sample_batch(256) from a full 10^6 "
                     "ring + SAC1 update), ~%.0f s each, 1 process, torch threads=1; combined at a_l_ratio=%g like the GPU leg"
                     % (n_env, n_upd, budget_s, a_l_ratio)}
    allc = {"value": combine(t_upd_all), "unit": "env-steps/s", "cores": nt_all, "kind": "port",
            "updates_per_s": combine(t_upd_all) / a_l_ratio, "learner_only_updates_per_s": 1.0 / t_upd_all,
            "host_cores_available": cores,
            "sample": "same rollout leg (a Python loop: one core) + %d updates with torch threads=%d (best of a sweep over 8 / 32 / %d "
                      "threads, ~%.0f s each)" % (n_upd_all, nt_all, cores, budget_s / 3.0)}
    return one, allc


# ------------------------------------------------------------------------------------------------
def fill_replay(rb, capacity, seed):
    import numpy as np
    import torch
    rs = np.random.RandomState(seed)
    chunk = 1 << 17
    for s in range(0, capacity, chunk):
        n = min(chunk, capacity - s)
        o, o2 = rs.randn(n, 8).astype(np.float32), rs.randn(n, 8).astype(np.float32)
        a = rs.uniform(-1, 1, (n, 2)).astype(np.float32)
        r = rs.randn(n).astype(np.float32)
        d = (rs.rand(n) < 0.01).astype(np.float32)
        rb.store_batch(*(torch.from_numpy(x).cuda() for x in (o, a, r, o2, d)))


def timed(fn, reps, warm=3):
    """Mean seconds per call of fn(), HIP events on the current stream."""
    import torch
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def stage_measurements(args, opt, rb, roll, d):
    """The other rows of SURVEY §8(d), each against its own roof (algorithmic bytes / FLOPs per unit of work)."""
    import torch
    out = {}
    n_env = int(opt.num_envs)
    t_live = None
    if getattr(roll, "_versions", False):
        # the version store (exact per-env weight adoption): right after a pull up to min(n_envs, max_ep_len) + 1 versions are live
        # and the forward runs grouped by version; max_ep_len steps after the last pull every env has been through an episode end
        # and the plain launch on the one version is the same computation
        t_live = timed(lambda: roll.step(20), 10) / 20.0
        _, vstate = roll.actor.version_state(with_slots=False)
        for _ in range(int(opt.max_ep_len) // 50 + 2):
            roll.step(50)
    t = timed(lambda: roll.step(20), 20) / 20.0   # 20 vector steps per call: two launches per step, no host work in between
    pf = policy_flops(opt.obs_dim, opt.act_dim, opt.hidden_sizes[0], opt.hidden_sizes[1])
    out["rollout_only"] = {"env_steps_per_s": n_env / t, "us_per_vector_step": t * 1e6, "num_envs": n_env,
                           "flop_per_env_step": pf, "achieved_TFLOPs": n_env * pf / t / 1e12,
                           "frac_of_f32_mfma_peak": n_env * pf / t / 1e12 / PEAK_F32_MFMA_TFLOPS,
                           "what": "policy forward + env.step + store of %d envs per launch sequence (no learner)" % n_env}
    if t_live is not None:
        out["rollout_only_versions_live"] = {"env_steps_per_s": n_env / t_live, "us_per_vector_step": t_live * 1e6, "num_envs": n_env,
                                             "row_tiles": vstate["tiles"], "versions_in_use_at_last_pull": vstate["live"],
                                             "what": "the same vector step right after the training loop's pushes: every env acts on the version "
                                                     "it pulled at its own episode end (example/dsac.py:127-130), envs grouped by version on the "
                                                     "device, one row tile per 32 envs of a version"}
    try:   # config 4's rollout ranks step 8192 envs: both launches of a vector step are latency chains at 4096 (one workgroup round), so the
        # rate grows with the envs per step (profiles/r03_rollout_sweep.txt)
        from distributed_drl_amd.agent import HyperParameters
        from distributed_drl_amd.workers import RolloutDevice
        o8 = HyperParameters(num_workers=1)
        o8.__dict__.update(opt.__dict__)
        o8.num_envs = 8192
        rb8 = d.ReplayBufferSAC1(opt.obs_dim, opt.act_dim, 1 << 20, seed=1)
        roll8 = RolloutDevice(None, rb8, o8, worker_index=1)
        t8 = timed(lambda: roll8.step(20), 20) / 20.0
        out["rollout_only_8192_envs"] = {"env_steps_per_s": 8192 / t8, "us_per_vector_step": t8 * 1e6, "num_envs": 8192,
                                         "achieved_TFLOPs": 8192 * pf / t8 / 1e12, "frac_of_f32_mfma_peak": 8192 * pf / t8 / 1e12 / PEAK_F32_MFMA_TFLOPS}
        del roll8, rb8
    except Exception as e:  # noqa
        out["rollout_only_8192_envs"] = {"error": repr(e)[:200]}
    o, o2 = torch.randn(n_env, 8, device="cuda"), torch.randn(n_env, 8, device="cuda")
    a, r, dn = torch.rand(n_env, 2, device="cuda"), torch.randn(n_env, device="cuda"), torch.zeros(n_env, device="cuda")
    t = timed(lambda: rb.store_batch(o, a, r, o2, dn), 200)
    out["store"] = {"n": n_env, "us": t * 1e6, "bytes_per_transition": 160, "GBps": n_env * 160 / t / 1e9,
                    "frac_of_hbm_peak": n_env * 160 / t / 1e9 / PEAK_HBM_GBPS, "transitions_per_s": n_env / t}
    B = int(opt.batch_size)
    t = timed(lambda: rb.sample_batch_device(B), 400)
    out["sample"] = {"batch": B, "us": t * 1e6, "bytes_per_batch": B * 164, "GBps": B * 164 / t / 1e9,
                     "frac_of_hbm_peak": B * 164 / t / 1e9 / PEAK_HBM_GBPS,
                     "note": "stand-alone launch (latency-bound); inside the learner loop the sampler rides in a forward launch"}
    K = 1024
    blk = torch.empty(K * B * 20, dtype=torch.float32, device="cuda")
    t = timed(lambda: rb.sample_many(B, K, blk), 10)
    out["sample_block"] = {"batch": B, "batches": K, "ms": t * 1e3, "bytes": K * B * 164, "GBps": K * B * 164 / t / 1e9,
                           "frac_of_hbm_peak": K * B * 164 / t / 1e9 / PEAK_HBM_GBPS,
                           "what": "the block of batches a shard owner draws for one step of a remote learner (configs 3/4): "
                                   "one sequential MT19937 index draw + one gather sweep"}
    try:   # the reference's num_learners > 1 (example/dsac.py:233: unsynchronised learners, last writer wins) on ONE GPU
        from distributed_drl_amd.workers import TrainDevice
        streams = [torch.cuda.Stream() for _ in range(2)]
        tds = []
        for i in range(2):
            ri = d.ReplayBufferSAC1(8, 2, 200000, seed=11 + i)
            fill_replay(ri, 200000, 500 + i)
            tds.append(TrainDevice(None, ri, opt, learner_index=10 + i, updates_per_graph=16))
        torch.cuda.synchronize()

        def both(n):
            for td, st in zip(tds, streams):
                with torch.cuda.stream(st):
                    td.run(n)
        both(64)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        both(3200)
        torch.cuda.synchronize()
        t = time.perf_counter() - t0
        out["two_learners_one_gpu"] = {"updates_per_s_total": 2 * 3200 / t, "us_per_update_each": t / 3200 * 1e6,
                                       "what": "two independent SAC1 learner loops (own ring, own stream, 16 updates per graph) running "
                                               "concurrently on the one GPU: the update's launches are latency chains at 1-2 workgroups per "
                                               "CU, a second learner fills the gaps"}
        del tds
    except Exception as e:  # noqa
        out["two_learners_one_gpu"] = {"error": repr(e)[:200]}
    try:   # example/dsac.py's own algorithm (SAC-v: policy + twin Q + V + target V, batch 100, hidden 300 x 2, lr 1e-3)
        from distributed_drl_amd.agent import Model

        class A:
            pass
        A.obs_dim, A.act_dim, A.ac_kwargs, A.gamma, A.polyak, A.lr, A.alpha, A.batch_size, A.seed, A.max_ep_len = \
            8, 2, dict(hidden_sizes=[300, 300]), 0.99, 0.995, 1e-3, 0.2, 100, 0, 1000
        m = Model(A)
        bv = {k: torch.randn(100, w, device="cuda").squeeze(-1) if w > 1 else torch.randn(100, device="cuda")
              for k, w in (("obs1", 8), ("obs2", 8), ("acts", 2), ("rews", 1), ("done", 1))}
        bv["done"] = (torch.rand(100, device="cuda") < 0.01).float()
        t = timed(lambda: m.train_device(bv), 300)
        out["sac_v_update"] = {"batch": 100, "hidden": [300, 300], "us": t * 1e6, "updates_per_s": 1.0 / t,
                               "what": "one Model.train step of example/model.py's SAC-v on a device batch, eager launches "
                                       "(direct-operand kernels, five launches; the 100 rows are padded to four 32-row tiles)"}
        from distributed_drl_amd.partition import _Loop
        rbv = d.ReplayBufferSAC1(8, 2, 100000, seed=3)
        fill_replay(rbv, 100000, 77)
        lp = _Loop(m, rbv, 16)
        lp.run(64)
        t = timed(lambda: lp.run(320), 5, warm=1) / 320
        out["sac_v_update"].update({"loop_us": t * 1e6, "loop_updates_per_s": 1.0 / t,
                                    "loop_what": "sample_batch(100) + train inside the graph-captured device loop (16 updates per graph)"})
        del lp, m, rbv
    except Exception as e:  # noqa
        out["sac_v_update"] = {"error": repr(e)[:200]}
    try:   # the reference's own call shapes: host NumPy in / out, one transition per store (PCIe-inclusive; never `value`)
        import numpy as np
        from distributed_drl_amd.agent import Learner
        rbh = d.ReplayBufferSAC1(8, 2, 100000, seed=5)
        rs = np.random.RandomState(0)
        oo, aa = rs.randn(8), rs.uniform(-1, 1, 2).astype(np.float32)
        for _ in range(200):
            rbh.store(oo, aa, 0.5, oo, False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(2000):
            rbh.store(oo, aa, 0.5, oo, False)                  # example/dsac.py:111: one transition, five host arrays
        torch.cuda.synchronize()
        t_store = (time.perf_counter() - t0) / 2000
        fill_replay(rbh, 100000, 9)
        t0 = time.perf_counter()
        for _ in range(300):
            bh = rbh.sample_batch(B)                            # dict of fresh float32 NumPy arrays (device -> host)
        t_samp = (time.perf_counter() - t0) / 300
        lh = Learner(opt, job="learner", index=77)
        for _ in range(5):
            lh.train(bh)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(300):
            lh.train(rbh.sample_batch(B))                       # worker_train's loop body through the host surface
        torch.cuda.synchronize()
        t_iter = (time.perf_counter() - t0) / 300
        rbh.prefetch(B, own_stream=True)                        # the reference's Cache (algos/sac1/sac1.py:103-130) inside the buffer
        for _ in range(20):
            lh.train(rbh.sample_batch(B))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(1000):
            lh.train(rbh.sample_batch(B))
        torch.cuda.synchronize()
        t_iter_pf = (time.perf_counter() - t0) / 1000
        rbh.prefetch(0)
        from distributed_drl_amd.agent import Actor
        ah = Actor(opt, job="worker", index=78)
        for _ in range(20):
            ah.get_action(oo)
        t0 = time.perf_counter()
        for _ in range(500):
            ah.get_action(oo)                                   # example/dsac.py:100: one observation up, one action down
        t_act = (time.perf_counter() - t0) / 500
        del ah
        out["host_surface_pcie_inclusive"] = {
            "store_per_s": 1.0 / t_store, "store_us": t_store * 1e6, "sample_batch_per_s": 1.0 / t_samp, "sample_batch_us": t_samp * 1e6,
            "sample_plus_train_per_s": 1.0 / t_iter, "sample_plus_train_us": t_iter * 1e6, "get_action_us": t_act * 1e6,
            "sample_plus_train_prefetch_per_s": 1.0 / t_iter_pf, "sample_plus_train_prefetch_us": t_iter_pf * 1e6,
            "what": "the reference's call shapes with HOST buffers crossing PCIe on every call: ReplayBuffer.store(obs, act, rew, next_obs, done) "
                    "of one transition (five NumPy values up), sample_batch(256) (a dict of NumPy arrays down), and worker_train's loop body "
                    "train(sample_batch()) (batch down and up again) — plain, and with the reference's Cache prefetch inside the buffer "
                    "(ReplayBuffer.prefetch(B, own_stream=True): ten draws in flight on the buffer's own stream; same batches, same order).  "
                    "Reported beside the device-resident figures, never as `value`."}
        del rbh, lh
    except Exception as e:  # noqa
        out["host_surface_pcie_inclusive"] = {"error": repr(e)[:200]}
    if args.cfg5_capacity != 0:
        out.update(config5_stages(args, d))
    return out


def config5_stages(args, d):
    """config 5: DQN (algos/dqn) on 84x84x4 float32 observations, batch 512 — the ring, its sampler and its learner at single-GPU scale.
    The learner legs do not depend on the ring (a box that cannot hold it still reports them); the ring is as large as the free HBM
    allows (--cfg5-capacity < 0: auto; the full 4 M transitions = 903 GB need shards)."""
    import torch
    from distributed_drl_amd import _lib, dqn
    out = {}
    obs_dim, B5 = 84 * 84 * 4, 512
    T5 = 4 * (2 * obs_dim + 1 + 2)

    class O5L:
        obs_dim, act_dim, hidden_size, gamma, lr, polyak, batch_size, seed = 84 * 84 * 4, 4, [400, 300], 0.99, 1e-3, 0.995, 512, 2
    g = torch.Generator(device="cuda").manual_seed(0)
    l5 = None
    try:
        l5 = dqn.Learner(O5L, "learner")
        names, vals = l5.get_weights()
        l5.set_weights(names[:1], [vals[0] / 64.0])      # 0..255 pixel inputs: keep layer 1 in range
        b5 = {"obs1": torch.randint(0, 256, (B5, obs_dim), device="cuda", generator=g).float(),
              "obs2": torch.randint(0, 256, (B5, obs_dim), device="cuda", generator=g).float(),
              "acts": torch.randint(0, 4, (B5,), device="cuda", generator=g).float(), "rews": torch.randn(B5, device="cuda", generator=g),
              "done": torch.zeros(B5, device="cuda")}
        t_u = timed(lambda: l5.train(b5, 0), 20, warm=3)
        st = l5.stage_times(b5, reps=20)
        h1, h2, A = 400, 300, 4
        # algorithmic FLOPs per launch group: three forwards (main @ x, main @ x2, target @ x2); backward of main @ x
        fl = {"layer1_forward": 2.0 * B5 * obs_dim * h1 * 3, "layer2_forward": 2.0 * B5 * h1 * h2 * 3, "head": 2.0 * B5 * h2 * A * 4,
              "layer2_backward": 2.0 * B5 * h1 * h2 * 2 + 2.0 * B5 * h2 * A, "layer1_wgrad": 2.0 * B5 * (obs_dim + 1) * h1}
        n_par = obs_dim * h1 + h1 + h1 * h2 + h2 + h2 * A + A
        adam_bytes = 40.0 * n_par                      # 28 B/param optimizer + 12 B/param target traffic
        names_k = {"layer1_forward": "k_wide<true> (504 workgroups, split-K) + k_wide_reduce", "layer2_forward": "k_gemm",
                   "head": "k_dqn_head (Q of the three evaluations, backup / loss / dQ, head dgrad: one launch)",
                   "layer2_backward": "k_gemm (layer-2 dgrad + wgrad, head wgrad)", "layer1_wgrad": "k_wide<false>"}
        order = ("stage", "layer1_forward", "layer2_forward", "head", "unused4", "unused5", "layer2_backward", "layer1_wgrad", "adam_polyak")
        per = {}
        for k, ms in zip(order, st):
            if k.startswith("unused"):
                continue
            e = {"us": ms * 1e3}
            if k in fl:
                e.update({"kernel": names_k[k], "flop": fl[k], "achieved_TFLOPs": fl[k] / (ms * 1e-3) / 1e12,
                          "frac": fl[k] / (ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, "bound": "mfma"})
            elif k == "adam_polyak":
                e.update({"kernel": "k_adam_polyak (flat, 11.4 M parameters)", "bytes": adam_bytes, "achieved_GBps": adam_bytes / (ms * 1e-3) / 1e9,
                          "frac": adam_bytes / (ms * 1e-3) / 1e9 / PEAK_HBM_GBPS, "bound": "hbm"})
            per[k] = e
        tot_fl = sum(fl.values())
        out["ddqn_update_cfg5"] = {
            "batch": 512, "obs_dim": obs_dim, "ms": t_u * 1e3, "updates_per_s": 1.0 / t_u,
            "what": "one Double-DQN update at config 5's learner shape on a device batch, eager: layer 1 (K = 28 224) on the LDS-DMA tiles of "
                    "csrc/wide_l1.h, layer 2 on k_gemm, the head in one launch (k_dqn_head), flat Adam + polyak",
            "roofline": {"kernel": "layer-1 forward (dominant): " + names_k["layer1_forward"], "bound": "mfma",
                         "achieved": per["layer1_forward"]["achieved_TFLOPs"], "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": per["layer1_forward"]["frac"], "traffic": None,
                         "update_flops": tot_fl, "update_achieved_TFLOPs": tot_fl / t_u / 1e12,
                         "update_frac": tot_fl / t_u / 1e12 / PEAK_F32_MFMA_TFLOPS,
                         "per_launch_group": per,
                         "how": "HIP events between the launch groups of 20 eager updates on the launch stream (ddrl_dqn_step_timed); rocprofv3 "
                                "summary of the same shape under profiles/"}}
    except Exception as e:  # noqa
        out["ddqn_update_cfg5"] = {"error": repr(e)[:300]}
    rb5 = None
    try:
        free_b, _ = torch.cuda.mem_get_info()
        cap = int(args.cfg5_capacity) if args.cfg5_capacity > 0 else int(min(1.15e6, 0.90 * free_b / T5))
        cap = max(2048, cap // 2048 * 2048)

        class O5:
            pass
        O5.obs_dim, O5.buffer_size, O5.batch_size, O5.save_dir = obs_dim, cap, B5, "."
        while True:
            try:
                rb5 = d.ReplayBufferDQN(O5, 0, seed=0)
                break
            except Exception:  # noqa  (out of memory: halve and retry; the stage still reports what it ran on)
                if cap <= 4096:
                    raise
                cap = max(4096, cap // 2 // 2048 * 2048)
                O5.buffer_size = cap
                torch.cuda.empty_cache()
        z = torch.zeros(2048, device="cuda")
        for s0 in range(0, cap, 2048):
            n = min(2048, cap - s0)
            x = torch.randint(0, 256, (n, obs_dim), device="cuda", generator=g).float()
            rb5.store_batch(x, z[:n], z[:n], x, z[:n])
        del x
        t_g = timed(lambda: rb5.sample_batch_device(B5), 64)
        out["config5_gather"] = {"batch": B5, "bytes_per_transition": T5, "ring_GB": cap * T5 / 1e9, "ring_transitions": cap,
                                 "us": t_g * 1e6, "bytes_per_batch": B5 * (2 * T5 + 4), "GBps": B5 * (2 * T5 + 4) / t_g / 1e9,
                                 "frac_of_hbm_peak": B5 * (2 * T5 + 4) / t_g / 1e9 / PEAK_HBM_GBPS,
                                 "what": "sample_batch(512) of 84x84x4 float32 transitions (MT19937 indices + five gathers) out of the largest ring "
                                         "the free HBM holds; the observation arrays are %.1f GB each: byte offsets far beyond 2^32, a ring far "
                                         "beyond the 256 MiB Infinity Cache" % (cap * obs_dim * 4 / 1e9)}
        if l5 is not None:
            def iteration():   # worker_train's loop body on config 5: sample_batch(512) -> train (algos/dqn/train.py:66-76, actor_learner.py)
                bb = rb5.sample_batch_device(B5)
                l5.train(bb, 0)
            t_i = timed(iteration, 20, warm=3)
            out["config5_learner_iteration"] = {"ms": t_i * 1e3, "iterations_per_s": 1.0 / t_i, "sample_ms": t_g * 1e3, "update_ms": t_u * 1e3,
                                                "sample_share": t_g / t_i,
                                                "what": "sample_batch(512) from the %.0f GB ring + one Double-DQN update, back to back on one stream: the "
                                                        "K = 28 224 layer-1 GEMMs (MFMA-bound), not the 231 MB gather (HBM-bound), set the rate" % (cap * T5 / 1e9)}
            t_f = timed(lambda: l5.train_from(rb5, 0), 20, warm=3)
            out["config5_learner_iteration_fused"] = {"ms": t_f * 1e3, "iterations_per_s": 1.0 / t_f, "vs_two_calls": t_f / t_i,
                                                      "what": "the same iteration as ONE call (ddrl_dqn_step_ring): indices drawn on the ring's stream, acts / rews / done "
                                                              "gathered, the layer-1 forward's LDS-DMA loads reading the sampled observation rows straight out of the "
                                                              "ring through the index list, only obs1 (for the weight gradient) gathered: 58 MB of the 231 MB batch; "
                                                              "bit-identical results (tests/test_gpu_math_fixtures.py)"}
    except Exception as e:  # noqa  (an out-of-memory box must not lose the headline line)
        out["config5_gather"] = dict(out.get("config5_gather", {}), error=str(e)[:200])
    del rb5
    torch.cuda.empty_cache()
    # config 5 at its STATED capacity: 4 M transitions on the one GPU as the opt-in compact ring (uint8 observations behind the float32
    # surface, lossless for integer-valued pixels; algos/dqn/train.py:43-52 x 4 M = 903 GB as float32, 226 GB compact)
    rbc = None
    try:
        free_b, _ = torch.cuda.mem_get_info()
        capc = 4 * 10 ** 6
        if args.cfg5_capacity > 0 or free_b < 2 * capc * obs_dim + 12 * capc + (4 << 30):
            capc = max(4096, min(capc, int(args.cfg5_capacity) if args.cfg5_capacity > 0 else int(0.85 * free_b / (2 * obs_dim + 12))) // 2048 * 2048)

        class O5C:
            pass
        O5C.obs_dim, O5C.buffer_size, O5C.batch_size, O5C.save_dir = obs_dim, capc, B5, "."
        rbc = d.ReplayBufferDQN(O5C, 0, seed=0, compact_obs=True)
        z = torch.zeros(2048, device="cuda")
        t0 = time.perf_counter()
        for s0 in range(0, capc, 2048):
            n = min(2048, capc - s0)
            x = torch.randint(0, 256, (n, obs_dim), device="cuda", generator=g).float()
            rbc.store_batch(x, z[:n], z[:n], x, z[:n])
        del x
        rbc.check()
        torch.cuda.synchronize()
        t_fill = time.perf_counter() - t0
        t_gc = timed(lambda: rbc.sample_batch_device(B5), 64)
        moved = B5 * (2 * obs_dim + 12 + 2 * 4 * obs_dim + 12 + 4)      # uint8 rows + three scalars read, float32 batch written, index read
        out["config5_gather_compact"] = {"batch": B5, "ring_transitions": capc, "ring_GB": capc * (2 * obs_dim + 12) / 1e9, "us": t_gc * 1e6,
                                         "fill_s": t_fill, "bytes_moved_per_batch": moved, "GBps": moved / t_gc / 1e9,
                                         "frac_of_hbm_peak": moved / t_gc / 1e9 / PEAK_HBM_GBPS,
                                         "delivered_float32_GBps": B5 * (2 * T5 + 4) / t_gc / 1e9,
                                         "what": "sample_batch(512) out of the 4 M-transition compact ring (uint8 observations, float32 batch): "
                                                 "config 5's stated replay capacity on one MI355X; results bit-identical to the float32 ring's on "
                                                 "integer-valued pixels (tests/test_gpu_replay.py)"}
        if l5 is not None:
            def iteration_c():
                l5.train(rbc.sample_batch_device(B5), 0)
            t_ic = timed(iteration_c, 20, warm=3)
            out["config5_learner_iteration_compact"] = {"ms": t_ic * 1e3, "iterations_per_s": 1.0 / t_ic, "sample_ms": t_gc * 1e3}
    except Exception as e:  # noqa
        out["config5_gather_compact"] = {"error": str(e)[:200]}
    del rbc, l5
    torch.cuda.empty_cache()
    return out


def free_running_measure(roll, trainer, opt, args, torch):
    """example/dsac.py:229-236 at N = 1: the rollout free-running on its own stream BESIDE the learner's graph loop on another
    (workers.FreeRunningLoop: no gate, transitions and weights cross at segment boundaries).  Both rates come from ONE timed region.
    The segment is `n` updates; the vector steps per segment are calibrated once so that both streams are busy for about the same
    time (neither half waits long at a boundary)."""
    from distributed_drl_amd.workers import FreeRunningLoop
    n = max(2, 2 * int(args.updates_per_graph))
    k_max = max(1, min(512, trainer.rb.max_size // int(opt.num_envs)))
    k = min(k_max, n)
    loop = FreeRunningLoop(roll, trainer, opt, steps_per_segment=k, updates_per_segment=n, timing=True)
    loop.run(6)
    t = loop.segment_times(last=4)
    loop.close()
    k2 = int(max(4, min(k_max, round(k * t["learner_ms"] / max(t["rollout_ms"], 1e-6)))))
    calib = {"steps_per_segment_tried": k, "rollout_ms": t["rollout_ms"], "learner_ms": t["learner_ms"]}
    loop = FreeRunningLoop(roll, trainer, opt, steps_per_segment=k2, updates_per_segment=n, timing=True)
    loop.run(4)
    loop.drain()
    segs = int(args.free_segments)
    e0, u0 = loop.env_steps, loop.updates
    t0 = time.perf_counter()
    loop.run(segs)
    loop.drain()
    dt = time.perf_counter() - t0
    ph = loop.segment_times(last=min(segs, 16))
    res = {"env_steps_per_s": (loop.env_steps - e0) / dt, "updates_per_s": (loop.updates - u0) / dt, "segments": segs, "seconds": dt,
           "steps_per_segment": k2, "updates_per_segment": n, "calibration": calib, "phase_ms_per_segment": ph,
           "rollout_stream_busy": ph["rollout_ms"] / (dt / segs * 1e3), "learner_stream_busy": (ph["learner_ms"] + ph["commit_ms"]) / (dt / segs * 1e3),
           "what": "workers.FreeRunningLoop: %d vector steps of %d envs on the rollout stream beside %d graph-loop updates on the learner "
                   "stream per segment; the rollout's transitions reach the replay ring by one commit per segment (staging ring -> "
                   "store_batch on the learner's stream), the learner's pushes reach the envs at the next boundary (per-env adoption at "
                   "episode ends as always); no actor/learner gate (example/dsac.py:76-150)" % (k2, int(opt.num_envs), n)}
    loop.close()
    return res


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))

    # the ONE JSON line goes to the real stdout; everything else this process (or a library under it: gloo and RCCL print
    # connection banners from C++) writes to fd 1 is sent to stderr
    json_fd = os.dup(1)
    sys.stdout.flush()
    os.dup2(2, 1)
    import numpy as np
    import torch
    import distributed_drl_amd as d
    from distributed_drl_amd import _lib, comm, partition
    from distributed_drl_amd.agent import HyperParameters, Learner
    from distributed_drl_amd.workers import RolloutDevice, TrainDevice

    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    cpu_one = cpu_all = None
    if world_env == 1 and not args.no_cpu_baseline:
        cpu_one, cpu_all = cpu_baseline(args.a_l_ratio, args.cpu_budget)   # before the GPU is touched: the GPU legs run last

    rank, world, local = comm.init_from_env()
    ndev = max(1, torch.cuda.device_count())
    torch.cuda.set_device(local % ndev)
    _lib.require_gpu()
    dev = torch.device("cuda", torch.cuda.current_device())
    roles = partition.Roles(world, rank, args.num_learners)
    dedicated = world > 1 and not (set(roles.learners) & set(roles.rollouts))
    num_envs = args.num_envs if args.num_envs is not None else (8192 if dedicated else 4096)

    opt = HyperParameters(num_workers=1, a_l_ratio=args.a_l_ratio)
    opt.num_envs, opt.batch_size = num_envs, args.batch
    opt.start_steps = -1          # policy phase from the first step (the expensive branch)
    opt.max_ep_len = 1000
    opt.seed = 0
    env_steps_per_step = num_envs * len(roles.rollouts)
    # per learner rank and step.  hold: the batches the reference's gate owes for one step's env steps, shared over the learner ranks
    # (algos/sac1/sac1.py:203-207); free: the config-2 figure at every N — example/dsac.py's workers, which have no gate at all
    # (example/dsac.py:76-150: rollouts and learners run unthrottled), with per-rank work fixed as N grows
    upd_by_gate = {"hold": max(1, int(round(env_steps_per_step / args.a_l_ratio / len(roles.learners)))),
                   "free": max(1, int(round(4096 / args.a_l_ratio)))}
    updates_per_step = upd_by_gate[args.gate]
    other_gate = "free" if args.gate == "hold" else "hold"
    shard_cap = args.capacity // len(roles.shard_owner)
    cfgd = dict(B=opt.batch_size, obs=opt.obs_dim, act=opt.act_dim, h1=opt.hidden_sizes[0], h2=opt.hidden_sizes[1])

    run = trainer = roll = rb = None
    if world == 1:
        rb = d.ReplayBufferSAC1(opt.obs_dim, opt.act_dim, args.capacity, seed=0)
        fill_replay(rb, args.capacity, 1234)
        trainer = TrainDevice(None, rb, opt, learner_index=0, updates_per_graph=args.updates_per_graph)
        keys, values = trainer.agent.get_weights()
        ps = d.ParameterServer(keys, values)
        trainer.ps = ps
        roll = RolloutDevice(ps, rb, opt, worker_index=0)

        def one_step(n_upd=updates_per_step):
            roll.step()
            trainer.run(n_upd)
    else:
        def make_replay():
            r = d.ReplayBufferSAC1(opt.obs_dim, opt.act_dim, shard_cap, seed=1000 + rank)
            fill_replay(r, shard_cap, 1234 + rank)
            return r
        def make_run(free_steps):
            return partition.PartitionedRun(opt, roles, make_replay, lambda r_: RolloutDevice(None, r_, opt, worker_index=rank),
                                            lambda: Learner(opt, job="learner", index=rank), seed=opt.seed, updates_per_graph=args.updates_per_graph,
                                            free_steps=free_steps)
        # --gate free at N > 1 is the free-running mode (partition.py): K vector steps per rollout rank beside the learners' updates,
        # the next step's blocks in flight under this step's updates, pushes adopted a step later off a communication stream
        run = make_run(args.free_steps if args.gate == "free" else 0)

        def one_step(n_upd=updates_per_step):
            run.step(n_upd)

    busy = [0.0]

    def block(n_upd=updates_per_step, one_step=one_step):
        torch.cuda.synchronize()
        comm.barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            one_step(n_upd)
        torch.cuda.synchronize()
        busy[0] = time.perf_counter() - t0             # this rank's own work (a rollout rank then waits for the learners)
        comm.barrier()
        return comm.allreduce_max(time.perf_counter() - t0, device=dev)

    for _ in range(args.warmup):
        one_step()
    t_gpu0 = time.perf_counter()
    dt = block()                                   # THE timed region: exactly --steps steps
    blocks = [dt]
    while True:   # every rank takes part in the decision (one collective per round)
        more = time.perf_counter() - t_gpu0 < args.gpu_seconds and len(blocks) < 50
        if comm.allreduce_max(1.0 if more else 0.0, device=dev) < 0.5:
            break
        blocks.append(block())

    vsteps_by_gate = {"hold": 1, "free": args.free_steps if world > 1 else 1}   # vector steps per rollout rank and step
    env_steps = args.steps * num_envs * len(roles.rollouts) * vsteps_by_gate[args.gate]
    updates = args.steps * updates_per_step * len(roles.learners)
    # the same timed block under the other gate setting (identical by construction when both owe the same updates per step: N = 1,
    # where the ungated figures come from workers.FreeRunningLoop instead).  N > 1: a second PartitionedRun in the other mode —
    # lock-step under the gate, free-running (partition.py: free_steps) without it
    run_other = None
    if world > 1:
        if run.learner is not None:
            run.check()
        run_other = make_run(args.free_steps if other_gate == "free" else 0)

        def other_step(n_upd):
            run_other.step(n_upd)
        for _ in range(max(1, args.warmup)):
            other_step(upd_by_gate[other_gate])
        dt_other = block(upd_by_gate[other_gate], other_step)
        if run_other.learner is not None:
            run_other.check()
    elif upd_by_gate[other_gate] != updates_per_step:
        one_step(upd_by_gate[other_gate])
        dt_other = block(upd_by_gate[other_gate])
    else:
        dt_other = dt
    # who ran where: (rank, device index, device UUID / PCI bus id, backend) gathered over the group
    props = torch.cuda.get_device_properties(dev)
    me = {"rank": rank, "device": dev.index, "uuid": str(getattr(props, "uuid", "")) or None,
          "pci_bus_id": getattr(props, "pci_bus_id", None), "name": props.name}
    rank_devices = [me]
    if world > 1:
        rank_devices = [None] * world
        torch.distributed.all_gather_object(rank_devices, me)
    rank_busy = role_times = learner_crc = None
    if run is not None and run.learner is not None:
        run.check()
    # series 2 of the N-GPU readout: what the rollout ranks produce when nothing holds them back — every rollout rank runs K vector steps
    # back to back (store into its local shard included, the policy versions of the run's pushes live), all ranks at once
    k_roll = 50
    my_roll = run.roll if run is not None else roll
    if my_roll is not None:
        for _ in range(5):
            my_roll.step()
    torch.cuda.synchronize()
    comm.barrier()
    t0 = time.perf_counter()
    if my_roll is not None:
        for _ in range(k_roll):
            my_roll.step()
    torch.cuda.synchronize()
    mine_roll = time.perf_counter() - t0
    comm.barrier()
    per_rank = [(k_roll * num_envs / mine_roll) if my_roll is not None else None]
    if world > 1:
        per_rank = [None] * world
        torch.distributed.all_gather_object(per_rank, (k_roll * num_envs / mine_roll) if my_roll is not None else None)
    roll_capacity = {"per_rank_env_steps_per_s": [None if v is None else round(v) for v in per_rank],
                     "sum_env_steps_per_s": float(sum(v for v in per_rank if v is not None)), "vector_steps": k_roll,
                     "what": "every rollout rank steps its %d envs %d times back to back, all ranks concurrently, no learner work: policy forward "
                             "(versions of the run's pushes live) + env.step + store into the local shard" % (num_envs, k_roll)}
    if run is not None:
        import zlib
        import torch.distributed as dist
        # data-parallel learners must hold the same parameters bit for bit after any number of steps (same start, same averaged gradient)
        learner_crc = [None] * world
        dist.all_gather_object(learner_crc, zlib.crc32(run.learner.get_weights_flat().cpu().numpy().tobytes()) if run.learner is not None else None)
        learner_crc = [c for c in learner_crc if c is not None]
        rank_busy = [None] * world
        dist.all_gather_object(rank_busy, busy[0])    # of the last repeated block (the same work as block 0, whose max over ranks is `dt`)
        # one extra block with the device drained at every phase boundary: where each role's time goes (diagnosis, not the metric)
        run.timing = True
        for k in [k for k in run.stats if k.startswith("s_")]:
            del run.stats[k]
        n_diag = max(1, min(args.steps, 3))
        for _ in range(n_diag):
            one_step()
        torch.cuda.synchronize()
        run.timing = False
        mine = {k: round(v / n_diag * 1e3, 4) for k, v in run.stats.items() if k.startswith("s_")}
        role_times = [None] * world
        dist.all_gather_object(role_times, mine)
    if rank != 0:
        return

    # ---- roofline of the update's launches: HIP events on the launch stream around learner-only graph replays -------
    lib = _lib.load()
    launches_per_update = 5
    gf = update_flops(**cfgd)
    roofline = {"kernel": "the %d launches of one SAC1 update (k_dfwd<0>, k_dfwd<1>, k_dg x3: layer 1 + fc GEMMs + heads / losses / "
                          "Adam / polyak in the epilogues), fp32 MFMA" % launches_per_update,
                "bound": "mfma", "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "update_flops": gf,
                "flops_per_launch": gf / launches_per_update}
    if trainer is not None:
        n_upd = 40 * args.updates_per_graph
        t_upd = timed(lambda: trainer.run(n_upd), 3, warm=1) / n_upd
        stage = np.zeros(_lib.SAC1_STAGES, np.float64)
        ms = ctypes.c_float()
        for st in (2, 5, 7, 8, 9):   # eager back-to-back launches of ONE stage (caches warm): a lower bound per launch
            _lib.check(lib.ddrl_sac1_stage_time(trainer.agent._h, st, args.stage_samples, ctypes.byref(ms), _lib.stream_ptr()))
            stage[st] = ms.value
        launch_s = t_upd / launches_per_update
        roofline.update({"achieved": gf / launches_per_update / launch_s / 1e12,   # per-launch FLOPs / per-launch time
                         "frac": gf / launches_per_update / launch_s / 1e12 / PEAK_F32_MFMA_TFLOPS,
                         "avg_launch_us": launch_s * 1e6, "update_us_learner_only": t_upd * 1e6,
                         "update_roofline_frac": (gf / (PEAK_F32_MFMA_TFLOPS * 1e12)) / t_upd,
                         "stage_us_warm": {k: round(float(stage[i]) * 1e3, 3) for k, i in
                                           (("k_dfwd<0>", 2), ("k_dfwd<1>", 5), ("k_dg bq", 7), ("k_dg mid", 8), ("k_dg pi", 9))}})
    else:
        upd_s = dt / (args.steps * updates_per_step)
        roofline.update({"achieved": gf / upd_s / 1e12, "frac": gf / upd_s / 1e12 / PEAK_F32_MFMA_TFLOPS,
                         "avg_launch_us": upd_s / launches_per_update * 1e6,
                         "note": "from the timed region of learner rank 0: env step + block transfers + the learner loop (%s)" %
                                 ("graph-captured, the sampler following the step's feed plan" if len(roles.learners) == 1 else
                                  "eager: gradients, RCCL all-reduce, Adam per update")})
    traffic, tsrc, stale = None, None, None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            traffic, tsrc = tj.get("bytes_per_launch_mean"), tj.get("source")
            stale = tj.get("kernel_source_sha256") != kernel_source_hash()
        except Exception:  # noqa
            pass
    roofline["traffic"] = traffic
    roofline["traffic_stale"] = stale   # true: csrc/ changed since the PMC passes behind profiles/traffic.json (tools/prof_round.sh + tools/make_traffic.py)
    roofline["traffic_source"] = ("static file profiles/traffic.json (%s); PMC counters cannot be collected inside this process" % tsrc) if traffic else None

    out = {
        "metric": "env-steps/s + learner updates/s, SAC1 LunarLanderContinuous-v2 @1/2/4/8 GPU",
        "value": env_steps / dt, "unit": "env-steps/s",
        "updates_per_s": updates / dt,                                    # sampled batches consumed per second, all learner ranks
        "optimizer_steps_per_s": args.steps * updates_per_step / dt,     # Adam steps of the learner GROUP: data-parallel learners make ONE step from their batches
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "SAC1 LunarLanderContinuous-v2 stand-in, %d vectorised envs per rollout rank + %d-transition device replay "
                               "(%d shard(s)), batch=%d, hidden (400,300), %d updates per learner rank per step, push every 300 updates" %
                               (num_envs, args.capacity, len(roles.shard_owner), args.batch, updates_per_step),
                   "num_envs": num_envs, "replay_capacity": args.capacity, "batch": args.batch,
                   "updates_per_step": updates_per_step, "updates_per_graph": args.updates_per_graph if len(roles.learners) == 1 else 0,
                   "gate": args.gate, "a_l_ratio": args.a_l_ratio,
                   "weak_scaling_read_against": "series.rollout_capacity_env_steps_per_s per rollout rank (env-steps half); "
                                                "series.learner_group_optimizer_steps_per_s per learner group (updates half)",
                   "weak_scaling_why": "BASELINE's configs fix the learner group (1 learner rank at N = 1 / 2 / 4, 2 data-parallel learner ranks of 8 at "
                                       "config 4), so the two halves of the metric scale differently BY CONSTRUCTION and the line carries them as "
                                       "three series. (1) series.rollout_capacity_env_steps_per_s: every rollout rank stepping its envs flat out, all "
                                       "ranks at once; efficiency = (capacity(N) / rollout_ranks(N)) / (capacity(1) / 1) — the series SURVEY 8(e) "
                                       "('env-steps/s is proportional to the rollout GPUs ... the >= 0.7 target is about not serialising on the "
                                       "learner') reads the north star's >= 0.7 against; nothing couples the rollout ranks but one 1.5 MB broadcast "
                                       "per 300 updates. (2) series.learner_group_optimizer_steps_per_s: ONE group at every N (SURVEY 8(e): 'updates/s "
                                       "is per learner group and does not scale with rollout GPUs'); what N changes is the feed (blocks from remote "
                                       "shards) and, at 2 learners, one all-reduce per update. (3) `value` keeps the actor/learner gate of "
                                       "algos/sac1/sac1.py:25,203-207 (env steps / sampled batches <= a_l_ratio): it equals a_l_ratio x the learner "
                                       "group's batches/s at every N, i.e. it follows series 2, and value(N) / (N value(1)) is bounded by "
                                       "learner_ranks / N (2/8 at config 4) whatever the kernels do. `value_ungated` (the same step without the gate, "
                                       "example/dsac.py:76-150) is reported but is NOT the read-out: its env steps per step grow 12x from N = 1 to 8 "
                                       "while the step stays as long as the learners' 2048 updates, so it reads 'efficiency' > 1 with the rollout "
                                       "ranks 99.9 % idle (env_steps_per_sample 12).",
                   "env_steps_per_sample": env_steps / max(1, updates), "learner_ranks": roles.learners, "rollout_ranks": roles.rollouts,
                   "backend": (torch.distributed.get_backend() if world > 1 else None), "devices": ndev,
                   "world_size": world, "rank_devices": rank_devices,
                   "distinct_devices": len({(r or {}).get("uuid") or (r or {}).get("pci_bus_id") or (r or {}).get("device") for r in rank_devices}),
                   "parallelism": roles.describe()},
        "repeat_blocks": {"n": len(blocks), "ms_per_step_median": float(np.median(blocks)) / args.steps * 1e3,
                          "ms_per_step_min": float(np.min(blocks)) / args.steps * 1e3, "ms_per_step_max": float(np.max(blocks)) / args.steps * 1e3,
                          "note": "`value` is block 0 (exactly --steps steps); the same block repeated to keep the GPU leg visible"},
        "roofline": roofline,
    }
    g_env = args.steps * num_envs * len(roles.rollouts) * vsteps_by_gate[other_gate]
    g_upd = args.steps * upd_by_gate[other_gate] * len(roles.learners)
    other = {"value": g_env / dt_other, "updates_per_s": g_upd / dt_other, "ms_per_step": dt_other / args.steps * 1e3,
             "updates_per_step": upd_by_gate[other_gate], "gate": other_gate, "vector_steps_per_rollout_rank_and_step": vsteps_by_gate[other_gate],
             "what": "the same timed block of --steps steps under --gate %s%s" % (other_gate, " (identical step at N = 1: not re-run)" if dt_other is dt else "")}
    out["value_ungated" if other_gate == "free" else "value_gated"] = other["value"]
    out["other_gate"] = other
    if args.gate == "free":
        out["value_ungated"] = out["value"]
    functional = os.environ.get("DDRL_BENCH_FUNCTIONAL_ONLY") == "1" or (world > 1 and out["config"]["distinct_devices"] < world) \
        or (world > 1 and out["config"]["backend"] != "nccl")
    out["functional_only"] = bool(functional)
    if functional:   # several ranks on one device (gloo, host-staged): the code path ran, the rate is not a scaling point
        out["functional_value"] = {"value": out["value"], "updates_per_s": out["updates_per_s"], "value_ungated": out.get("value_ungated"),
                                   "note": "ranks shared a GPU over gloo: host-staged transfers on one device — NOT an N-GPU measurement"}
        out["value"], out["updates_per_s"] = 0.0, 0.0
        out.pop("value_ungated", None)
    if run is not None:
        out["partition_stats"] = {k: v for k, v in run.stats.items() if not k.startswith("s_")}
        s_upd = [t.get("s_updates") for t in role_times if t and t.get("s_updates")]
        out["scaling_readout"] = {
            "env_steps_per_sample": env_steps / max(1, updates),
            "rank_busy_s_in_timed_block": [round(b, 4) for b in rank_busy],
            "roles": ["learner+rollout" if (r in roles.learners and r in roles.rollouts) else ("learner" if r in roles.learners else "rollout") for r in range(world)],
            "phase_ms_per_step_drained": role_times,
            "learner_only_updates_per_s": (updates_per_step * len(roles.learners) / (max(s_upd) * 1e-3)) if s_upd else None,
            "rollout_only_env_steps_per_s_per_rank": [round(num_envs / (t["s_env"] * 1e-3)) if t and t.get("s_env") else None for t in role_times],
            "note": "value = env steps of the timed block / its wall time (max over ranks); under --gate hold the learner rank(s) bound it. "
                    "phase times come from one extra block with the device drained at every phase boundary (s_env: vector env step, "
                    "s_serve: shard owner draws + sends its blocks, s_receive: learner posts receives + lays the plan, s_updates: the "
                    "updates, s_push: parameter broadcast, s_drain: sends complete)"}
    if learner_crc is not None:
        out["partition_stats"]["learner_weight_crc32"] = learner_crc
        out["partition_stats"]["learners_identical"] = len(set(learner_crc)) == 1
    lg_steps = None
    if world == 1 and "update_us_learner_only" in roofline:
        lg_steps = 1e6 / roofline["update_us_learner_only"]
    elif run is not None and out["scaling_readout"]["learner_only_updates_per_s"]:
        lg_steps = out["scaling_readout"]["learner_only_updates_per_s"] / len(roles.learners)
    out["series"] = {
        "gated_env_steps_per_s": out["value"] if not functional else out["functional_value"]["value"],
        "learner_group_optimizer_steps_per_s": lg_steps,
        "learner_group_batches_per_s": None if lg_steps is None else lg_steps * len(roles.learners),
        "rollout_capacity_env_steps_per_s": roll_capacity["sum_env_steps_per_s"],
        "rollout_capacity": roll_capacity,
        "learner_ranks": len(roles.learners), "rollout_ranks": len(roles.rollouts),
        "functional_only": bool(functional),
        "note": "learner-group rate: the learner rank(s) alone%s; rollout capacity: rollout-only block, all rollout ranks at once; gated: the "
                "timed region (`value`).  See config.weak_scaling_why for which series the 0.7 target is read against." %
                (" (graph loop, HIP events)" if world == 1 else " (updates phase of one drained block: feed attached, all-reduce included)")}
    if world == 1 and trainer is not None and not args.no_free:
        fr = free_running_measure(roll, trainer, opt, args, torch)
        out["free_running"] = fr
        # at N = 1 the gated step and the "free" step of rounds 2-5 were the same step; the ungated figures are now what the two halves
        # reach when NOTHING holds either back and they share the GPU
        out["value_ungated"] = fr["env_steps_per_s"]
        out["updates_per_s_ungated"] = fr["updates_per_s"]
        out["series"]["rollout_capacity_with_learner_env_steps_per_s"] = fr["env_steps_per_s"]
        out["series"]["learner_updates_per_s_beside_free_rollout"] = fr["updates_per_s"]
    if world == 1 and not args.no_stages:
        out["stages"] = stage_measurements(args, opt, rb, roll, d)
    if world == 1:
        out["scaling_readout"] = {
            "env_steps_per_sample": env_steps / max(1, updates), "roles": ["learner+rollout"],
            "learner_only_updates_per_s": (1e6 / roofline["update_us_learner_only"]) if "update_us_learner_only" in roofline else None,
            "rollout_only_env_steps_per_s_per_rank": [out.get("stages", {}).get("rollout_only", {}).get("env_steps_per_s")],
            "note": "one rank runs the envs and the learner back to back; under the gate the learner is 99.9 % of the step"}
    out["cpu_baseline"] = cpu_one
    out["cpu_baseline_all_cores"] = cpu_all
    if cpu_one is not None:
        out["gpu_over_cpu"] = {"env_steps": out["value"] / cpu_one["value"], "updates": out["updates_per_s"] / cpu_one["updates_per_s"]}
    sys.stdout.flush()
    os.write(json_fd, (json.dumps(out) + "\n").encode())


if __name__ == "__main__":
    main()
