#!/usr/bin/env python3
"""bench.py — env-steps/s + learner updates/s of the SAC1 actor-learner hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workload (BASELINE.json configs[1]): SAC1 on the LunarLanderContinuous-v2 stand-in, 4096
vectorised envs + 1M-transition device replay per GPU, batch 256, hidden (400, 300).
One "step" = one pass of the hot path over one batch of environments:
    rollout : batched policy forward (4096 x 8->400->300->(2,2)) -> env.step kernel -> store 4096
    learner : num_envs / a_l_ratio updates, each = MT19937 sample of 256 + gather -> SAC1 update
              (the reference's actor/learner gate keeps steps / sample_times <= a_l_ratio,
              algos/sac1/sac1.py:205; default a_l_ratio = 2, sac1.py:25)
    ps      : push of the flat parameter vector every 300 updates (sac1.py:149), pull by the actor
Synthetic data: the ring is pre-filled to capacity with seeded synthetic transitions (SURVEY
§8(d)); weights are glorot/zeros random init.  Inputs are resident in HBM when timing starts.

N > 1 (weak scaling, per-GPU work fixed): one process per GPU, each with its own envs, its own
replay shard (local store, local sampling — the per-node buffers of algos/dqn/train.py:392-411)
and its own learner (`num_learners` independent learners, example/dsac.py:233); the only exchange
is ps.push/pull = ONE RCCL broadcast of the flat parameter vector per push (source rotates:
deterministic last-writer-wins).  `--dp-learners` instead all-reduces the gradient every update
(BASELINE config 4 semantics).

Prints ONE JSON line on rank 0.  `value` = whole-job env-steps/s; `updates_per_s` rides along.
`roofline` prices the four MFMA stage launches of an update (k_fwd<0>, k_fwd<1>, k_bwdq, k_gemm: layer 1
on the matrix cores + the fc GEMMs + the fused head / loss / optimizer epilogues = all of an update's
network FLOPs) from HIP-event stage timings taken right after the timed region; `cpu_baseline` times the oracle
(CPU restatement of the reference path) on this box's host cores for a bounded sample.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak


def gemm_flops_per_update(B, obs, act, h1, h2):
    fwd = 8 * 2 * B * h1 * h2      # 8 network evaluations, layer 2
    dgrad = 4 * 2 * B * h2 * h1    # q1(x,a), q2(x,a), q1(x,pi), pi
    wgrad = 3 * 2 * h1 * B * h2    # q1, q2, pi
    return fwd + dgrad + wgrad


def update_flops(B, obs, act, h1, h2):
    """1 853 800 MAC per sample -> 949 145 600 FLOP at B = 256."""
    pol = obs * h1 + h1 * h2 + 2 * h2 * act
    q = (obs + act) * h1 + h1 * h2 + h2
    fwd = 3 * pol + 5 * q
    bwd = (2 * pol - obs * h1) + q + 2 * (2 * q - (obs + act) * h1)
    return 2 * B * (fwd + bwd)


def fill_replay(rb, capacity, seed):
    rs = np.random.RandomState(seed)
    chunk = 1 << 17
    for s in range(0, capacity, chunk):
        n = min(chunk, capacity - s)
        o, o2 = rs.randn(n, 8).astype(np.float32), rs.randn(n, 8).astype(np.float32)
        a = rs.uniform(-1, 1, (n, 2)).astype(np.float32)
        r = rs.randn(n).astype(np.float32)
        d = (rs.rand(n) < 0.01).astype(np.float32)
        rb.store_batch(*(torch.from_numpy(x).cuda() for x in (o, a, r, o2, d)))


def cpu_baseline(a_l_ratio, budget_s=10.0):
    """The oracle (CPU restatement of the reference path, kind "port") on the host cores, single
    process like Ray local mode, torch pinned to 1 thread like the reference's session config
    (algos/sac1/actor_learner.py:110-111)."""
    from oracle import sac1_oracle as so
    from oracle.env_oracle import LanderOracle
    from oracle.replay_oracle import ReplayBufferOracle
    threads = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        cfg = so.Config()
        params = so.init_params(cfg, 0)
        rb = ReplayBufferOracle(8, 2, 10 ** 6, seed=0)
        rs = np.random.RandomState(1234)
        n0 = 20000
        rb.obs1_buf[:n0], rb.obs2_buf[:n0] = rs.randn(n0, 8), rs.randn(n0, 8)
        rb.acts_buf[:n0], rb.rews_buf[:n0] = rs.uniform(-1, 1, (n0, 2)), rs.randn(n0)
        rb.ptr = rb.size = rb.steps = n0
        env = LanderOracle(1, seed=0)
        o = env.obs()
        rn = np.random.RandomState(1)
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
        # learner leg: sample_batch(256) + one SAC1 update
        learner = so.Sac1Oracle(cfg, params, torch.float32)
        t0, n_upd = time.perf_counter(), 0
        while time.perf_counter() - t0 < budget_s:
            batch = rb.sample_batch(256)
            eps = [rn.randn(256, 2).astype(np.float32) for _ in range(3)]
            learner.step(batch, *eps)
            n_upd += 1
        t_upd = (time.perf_counter() - t0) / n_upd
    finally:
        torch.set_num_threads(threads)
    env_rate = 1.0 / (t_env + t_upd / a_l_ratio)
    return {"value": env_rate, "unit": "env-steps/s", "cores": 1, "kind": "port",
            "updates_per_s": env_rate / a_l_ratio,
            "rollout_only_env_steps_per_s": 1.0 / t_env, "learner_only_updates_per_s": 1.0 / t_upd,
            "host_cores_available": os.cpu_count(),
            "sample": "%d env steps (batch-of-1 policy forward + env.step + store) and %d updates "
                      "(sample_batch(256) + SAC1 update), ~%.0f s each, 1 process, torch threads=1; "
                      "combined at a_l_ratio=%g like the GPU leg" % (n_env, n_upd, budget_s, a_l_ratio)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--num-envs", type=int, default=4096)
    ap.add_argument("--capacity", type=int, default=10 ** 6)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--a-l-ratio", type=float, default=2.0)
    ap.add_argument("--updates-per-graph", type=int, default=50)  # divides push_freq = 300: no eager remainder between pushes (32: -1.2 %)
    ap.add_argument("--dp-learners", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=10.0)
    ap.add_argument("--stage-samples", type=int, default=200)
    args = ap.parse_args()

    import distributed_drl_amd as d
    from distributed_drl_amd import _lib, comm
    from distributed_drl_amd.agent import HyperParameters
    from distributed_drl_amd.workers import RolloutDevice, TrainDevice

    rank, world, local = comm.init_from_env()
    if world != args.gpus and rank == 0:
        print("note: --gpus %d but WORLD_SIZE=%d; using WORLD_SIZE" % (args.gpus, world), file=sys.stderr)
    torch.cuda.set_device(local % torch.cuda.device_count())
    _lib.require_gpu()
    dev = torch.device("cuda", torch.cuda.current_device())

    opt = HyperParameters(num_workers=1, a_l_ratio=args.a_l_ratio)
    opt.num_envs, opt.batch_size = args.num_envs, args.batch
    opt.start_steps = -1          # policy phase from the first step (the expensive branch)
    opt.max_ep_len = 1000
    opt.seed = 0
    updates_per_step = max(1, int(round(args.num_envs / args.a_l_ratio)))

    rb = d.ReplayBufferSAC1(opt.obs_dim, opt.act_dim, args.capacity, seed=rank)
    fill_replay(rb, args.capacity, 1234 + rank)
    bcast = None
    push_no = [0]

    def on_push(flat):
        if bcast is not None:
            # last-writer-wins between the independent learners, in a fixed rotating order
            bcast.src = push_no[0] % world
            bcast.sync(flat)
            push_no[0] += 1

    trainer = TrainDevice(None, rb, opt, learner_index=rank, updates_per_graph=args.updates_per_graph, on_push=on_push)
    keys, values = trainer.agent.get_weights()
    ps = d.ParameterServer(keys, values)
    trainer.ps = ps
    if world > 1:
        bcast = comm.ParamBroadcast(trainer.agent.n_params, dev, src=0)
    roll = RolloutDevice(ps, rb, opt, worker_index=rank)
    if bcast is not None:
        n_pi = roll.actor.n_params
        orig_pull = roll.pull

        def pull_bcast():
            if bcast.version != getattr(roll, "_bv", -1):
                roll._bv = bcast.version
                roll.actor.set_weights_flat(bcast.buf[:n_pi])
                return True
            return orig_pull()
        roll.pull = pull_bcast

    dp_flat = None

    def one_step():
        roll.step()
        if args.dp_learners and world > 1:
            for _ in range(updates_per_step):   # synchronous data-parallel learners (config 4)
                batch = rb.sample_batch_device(opt.batch_size)
                g = trainer.agent.compute_gradients(batch)
                comm.allreduce_mean_(g)
                trainer.agent.apply_gradients(g)
        else:
            trainer.run(updates_per_step)

    for _ in range(args.warmup):
        one_step()
    torch.cuda.synchronize()
    comm.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    torch.cuda.synchronize()
    comm.barrier()
    dt = comm.allreduce_max(time.perf_counter() - t0, device=dev)

    env_steps = args.steps * args.num_envs * world
    updates = args.steps * updates_per_step * world
    if rank != 0:
        return

    # ---- roofline of the dominant kernel (k_gemm), HIP events on the launch stream -------------
    cfgd = dict(B=opt.batch_size, obs=opt.obs_dim, act=opt.act_dim, h1=opt.hidden_sizes[0], h2=opt.hidden_sizes[1])
    lib = _lib.load()
    bufs = (ctypes.c_void_p * 8)()
    _lib.check(lib.ddrl_sac1_input_buffers(trainer.agent._h, 0, bufs))
    stage = np.zeros(_lib.SAC1_STAGES, np.float64)
    ms = ctypes.c_float()
    for st in range(1, 11):   # idempotent stages, `stage_samples` back-to-back launches between two HIP events
        _lib.check(lib.ddrl_sac1_stage_time(trainer.agent._h, st, args.stage_samples, ctypes.byref(ms), _lib.stream_ptr()))
        stage[st] = ms.value
    gemm_ms = [stage[i] for i in (2, 5, 7, 9)]   # the four MFMA stage launches of one update
    gemm_launch_s = float(np.mean(gemm_ms)) * 1e-3
    # algorithmic FLOPs per launch = SURVEY §8(d)'s per-update figure / 4 launches (on the fused path the
    # launches also carry layer 1, the heads, the losses and the optimizer step: nothing else computes)
    gf = update_flops(**cfgd)
    achieved = gf / 4.0 / gemm_launch_s / 1e12
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get("k_gemm_bytes_per_launch")
        except Exception:
            traffic = None
    roofline = {"kernel": "MFMA stage launches k_fwd<0> / k_fwd<1> / k_bwdq / k_gemm (4 per update, fp32 MFMA)", "bound": "mfma",
                "achieved": achieved, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved / PEAK_F32_MFMA_TFLOPS, "traffic": traffic,
                "flops_per_launch": gf / 4.0, "avg_launch_us": gemm_launch_s * 1e6,
                "stage_us": [round(float(x) * 1e3, 3) for x in stage],
                "stages_sum_us": float(stage.sum() * 1e3),
                "update_flops": update_flops(**cfgd)}
    upd_us_in_loop = dt / (args.steps * updates_per_step) * 1e6  # includes the rollout share
    roofline["update_roofline_frac"] = (update_flops(**cfgd) / (PEAK_F32_MFMA_TFLOPS * 1e12)) / (upd_us_in_loop * 1e-6)

    out = {
        "metric": "env-steps/s + learner updates/s, SAC1 LunarLanderContinuous-v2 @1/2/4/8 GPU",
        "value": env_steps / dt, "unit": "env-steps/s",
        "updates_per_s": updates / dt,
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "SAC1 LunarLanderContinuous-v2 stand-in, %d vectorised envs + %d-transition "
                               "device replay per GPU, batch=%d, hidden (400,300), a_l_ratio=%g -> %d updates "
                               "per vector step, push every 300 updates" %
                               (args.num_envs, args.capacity, args.batch, args.a_l_ratio, updates_per_step),
                   "num_envs": args.num_envs, "replay_capacity": args.capacity, "batch": args.batch,
                   "updates_per_step": updates_per_step, "updates_per_graph": args.updates_per_graph,
                   "parallelism": ("dp-learners allreduce" if args.dp_learners else
                                   "replicas: per-GPU envs + replay shard + learner; ps.push/pull = RCCL broadcast")
                   if world > 1 else "single GPU"},
        "roofline": roofline,
    }
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.a_l_ratio, args.cpu_budget)
        out["gpu_over_cpu"] = {"env_steps": out["value"] / out["cpu_baseline"]["value"],
                               "updates": out["updates_per_s"] / out["cpu_baseline"]["updates_per_s"]}
    else:
        out["cpu_baseline"] = None
    print(json.dumps(out))


if __name__ == "__main__":
    main()
