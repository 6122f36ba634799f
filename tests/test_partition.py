"""Rank-role partitioning (BASELINE configs 3 / 4): role tables, the shared shard schedule, packed batches — host logic on
CPU; and, marked gpu, the world-size-2 run of the real HIP ring + learner on both ranks of ONE GPU (gloo transport)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_roles_match_the_baseline_configs():
    from distributed_drl_amd.partition import Roles
    r1 = Roles(1, 0)
    assert (r1.learners, r1.rollouts, r1.shard_owner, r1.my_shard) == ([0], [0], [0], 0)
    # config 3: learner on rank 0, envs + shard on both ranks
    a, b = Roles(2, 0), Roles(2, 1)
    assert a.learners == [0] and a.rollouts == [0, 1] and a.is_learner and a.is_rollout and a.my_shard == 0
    assert not b.is_learner and b.is_rollout and b.my_shard == 1
    # config 4: 2 learner ranks + 6 rollout ranks with a shard each
    rs = [Roles(8, r) for r in range(8)]
    assert rs[0].learners == [0, 1] and rs[0].rollouts == [2, 3, 4, 5, 6, 7] and rs[0].shard_owner == [2, 3, 4, 5, 6, 7]
    assert [r.is_learner for r in rs] == [True, True] + [False] * 6 and rs[1].my_shard is None and rs[5].my_shard == 3
    assert Roles(4, 0).learners == [0] and Roles(4, 3).rollouts == [1, 2, 3]
    assert "gradient all-reduce" in rs[0].describe()
    # sizes the partition cannot serve fail at construction with a clear message, not in the middle of a run
    for world, nl in ((2, 2), (3, 3), (24, None)):
        with pytest.raises(ValueError, match="rollout rank|shard owners"):
            Roles(world, 0, num_learners=nl)
    assert len(Roles(16, 0).shard_owner) == 12                  # 12 remote blocks per step <= MAX_FEED (16)


def test_schedule_is_the_reference_choice_stream_and_identical_on_every_rank():
    from distributed_drl_amd.partition import Roles, Schedule
    plans = []
    for rank in (0, 5):
        s = Schedule(Roles(8, rank), seed=3)
        plans.append([s.next() for _ in range(50)])
    assert plans[0] == plans[1]
    # learner l draws np.random.choice(num_buffers, 1)[0] on its own legacy MT19937 stream (sac_ray.py:137)
    for i, l in enumerate((0, 1)):
        np.random.seed(3 + 7919 * i)
        want = [2 + int(np.random.choice(6, 1)[0]) for _ in range(50)]
        assert [dict(p)[l] for p in plans[0]] == want


def test_packed_batch_views():
    from distributed_drl_amd.partition import batch_floats, batch_views
    n = batch_floats(8, 2, 256)
    assert n * 4 == 20480          # 20 480 B at B = 256
    flat = torch.arange(n, dtype=torch.float32)
    v = batch_views(flat, 8, 2, 256)
    assert v["obs1"].shape == (256, 8) and v["acts"].shape == (256, 2) and v["done"].shape == (256,)
    assert v["obs2"][0, 0] == 256 * 8 and v["rews"][0] == 256 * 18 and v["done"][-1] == n - 1
    v["rews"][3] = -1.0
    assert flat[256 * 18 + 3] == -1.0   # views, not copies


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _prefilled_shard(d, opt, r, cap=4096, n=500):
    """Shard r with n rows whose reward 1000 r + i shows the ring and row a sampled transition came from."""
    rb = d.ReplayBufferSAC1(opt.obs_dim, opt.act_dim, cap, seed=100 + r)
    rs = np.random.RandomState(r)
    rb.store_batch(*(torch.from_numpy(x).cuda() for x in (
        rs.randn(n, 8).astype(np.float32), rs.uniform(-1, 1, (n, 2)).astype(np.float32),
        (1000.0 * r + np.arange(n)).astype(np.float32), rs.randn(n, 8).astype(np.float32), np.zeros(n, np.float32))))
    return rb


def _rows(rb, idx):
    """The packed batch [obs1 | obs2 | acts | rews | done] the ring hands out for indices idx (host copy)."""
    g = rb.rings()
    return np.concatenate([g[k][torch.from_numpy(idx).cuda()].reshape(-1).cpu().numpy() for k in ("obs1_buf", "obs2_buf", "acts_buf", "rews_buf", "done_buf")])


# (envs per rollout rank, batch, shard capacity, rows pre-filled, updates) — the toy the first tests run, and BASELINE config 4's own sizes:
# 6 x 8192 envs, batch 256, 10^6 transitions over 6 shards (bench.py: capacity // shards), pre-filled so that the rings WRAP within the run
TOY = (64, 32, 4096, 500, 14)
CONFIG3 = (4096, 256, 10 ** 6 // 2, 480000, 10)
CONFIG4 = (8192, 256, 10 ** 6 // 6, 120000, 10)


def _gpu_worker(rank, world, port, q, num_learners, sizes=TOY):
    try:
        n_envs, batch, cap, prefill, n_upd = sizes
        sys.path.insert(0, ROOT)
        os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                          DDRL_DIST_BACKEND="gloo")
        import distributed_drl_amd as d
        import torch.distributed as dist
        from distributed_drl_amd import _lib, comm, partition
        from distributed_drl_amd.agent import HyperParameters, Learner
        from distributed_drl_amd.workers import RolloutDevice
        r, w, _ = comm.init_from_env()
        torch.cuda.set_device(0)
        _lib.require_gpu()
        opt = HyperParameters()
        opt.num_envs, opt.batch_size, opt.seed, opt.start_steps, opt.max_ep_len, opt.push_freq = n_envs, batch, 5, -1, 50, 6
        B = opt.batch_size
        roles = partition.Roles(w, r, num_learners=num_learners)
        run = partition.PartitionedRun(opt, roles, lambda: _prefilled_shard(d, opt, r, cap, prefill), lambda rb: RolloutDevice(None, rb, opt, worker_index=r),
                                       lambda: Learner(opt, job="learner", index=0), seed=9, updates_per_graph=0)
        pi0 = run.bcast.buf[: 8 * 400 + 400 + 400 * 300 + 300 + 2 * (300 * 2 + 2)].clone()
        if run.roll is not None:   # the learner's initial weights reached every rollout rank's actor
            assert torch.equal(run.roll.actor.get_weights_flat(), pi0)
        sched = partition.Schedule(roles, seed=9)
        plans = [sched.next() for _ in range(n_upd)]
        # What every ring must hand out, from NumPy's own legacy stream: the ring's sampler is np.random.seed(100 + r);
        # at step u it holds min(capacity, prefill + n_envs (u + 1)) rows (the env step of the step is stored before its updates) and
        # serves learner 0's batch before learner 1's.
        rs = np.random.RandomState(100 + r)
        handed, trained = [], []
        for u in range(n_upd):   # one update per step() call so that the batch of every update can be inspected
            run.step(1)
            torch.cuda.synchronize()
            if run.rb is not None:
                for l, owner in plans[u]:
                    if owner == r:
                        handed.append((u, l, _rows(run.rb, rs.randint(0, min(cap, prefill + opt.num_envs * (u + 1)), B))))
            if run.loop is not None:   # single learner: the device loop gathered update u's batch into input set u & 1
                v = run.learner.input_batch(u & 1)
                trained.append(torch.cat([v[k].reshape(-1) for k in ("obs1", "obs2", "acts", "rews", "done")]).cpu().numpy().copy())
            elif run.learner is not None:   # data-parallel learners: the input set the update just trained on
                v = run.learner.input_batch(run.learner._dp_last_set)
                trained.append(torch.cat([v[k].reshape(-1) for k in ("obs1", "obs2", "acts", "rews", "done")]).cpu().numpy().copy())
        everything = [None] * w
        dist.all_gather_object(everything, handed)
        if roles.is_learner:
            # update u trained on exactly the batch the scheduled owner's ring handed out for it (bit for bit; never a mix)
            want = {(u, l): rows for per_rank in everything for (u, l, rows) in per_rank}
            for u in range(n_upd):
                np.testing.assert_array_equal(trained[u], want[(u, r)], err_msg="update %d of learner %d" % (u, r))
            assert run.learner.opt_steps() == (n_upd, n_upd)
            mine = [dict(p)[r] for p in plans]
            assert run.stats["local_batches"] == mine.count(r) and run.stats["remote_batches"] == n_upd - mine.count(r)
        if run.rb is not None:
            # the shard's sampler advanced once per batch it served, the ring took its local stores
            served = sum(1 for p in plans for _, owner in p if owner == r)
            samples, steps, size = run.rb.get_counts()
            assert samples == served and steps == prefill + n_upd * opt.num_envs and size == min(cap, steps)
            assert run.stats["sent_batches"] == sum(1 for p in plans for l, owner in p if owner == r and l != r)
        # a push every 6 updates + the initial one: the rollout ranks run the learner's pushed policy
        assert run.stats["pushes"] == 1 + n_upd // 6
        flat = run.bcast.buf.clone()
        if roles.is_learner:
            assert not torch.equal(flat[: pi0.numel()], pi0)
        if run.roll is not None:
            assert torch.equal(run.roll.actor.get_weights_flat(), flat[: pi0.numel()])
        if num_learners == 2:   # synchronous data parallel: both learners hold the same parameters after every update
            ws = [None] * w
            dist.all_gather_object(ws, run.learner.get_weights_flat().cpu().numpy() if roles.is_learner else None)
            np.testing.assert_array_equal(ws[0], ws[1])
        comm.barrier()
        q.put((rank, "ok"))
    except Exception:  # noqa
        import traceback
        q.put((rank, "FAIL: " + traceback.format_exc()))


def _gpu_worker_async(rank, world, port, q, num_learners):
    """The same partitioned run twice — once drained after every step (the order the test above checks batch by batch),
    once with the host running whole steps ahead of the device (graph-captured loop, several updates per step, no
    synchronisation until the end): identical parameters, counters and sampler state, i.e. no staging buffer, plan or
    region was rewritten under a step still in flight."""
    try:
        sys.path.insert(0, ROOT)
        os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                          DDRL_DIST_BACKEND="gloo")
        import distributed_drl_amd as d
        from distributed_drl_amd import _lib, comm, partition
        from distributed_drl_amd.agent import HyperParameters, Learner
        from distributed_drl_amd.workers import RolloutDevice
        r, w, _ = comm.init_from_env()
        torch.cuda.set_device(0)
        _lib.require_gpu()
        opt = HyperParameters()
        opt.num_envs, opt.batch_size, opt.seed, opt.start_steps, opt.max_ep_len, opt.push_freq = 64, 32, 5, -1, 50, 7
        roles = partition.Roles(w, r, num_learners=num_learners)
        n_steps, per_step = 9, 5        # 45 updates: pushes at 7, 14, ... fall inside steps; 9 steps > PLAN_STAGES
        if os.environ.get("DIAG_TRACE") == "2":
            n_steps, per_step = 45, 1
        per_graph = int(os.environ.get("DIAG_PER_GRAPH", "2"))
        opt.push_freq = int(os.environ.get("DIAG_PUSH_FREQ", "7"))

        def one(drain):
            run = partition.PartitionedRun(opt, roles, lambda: _prefilled_shard(d, opt, r), lambda rb: RolloutDevice(None, rb, opt, worker_index=r),
                                           lambda: Learner(opt, job="learner", index=0), seed=9, updates_per_graph=per_graph)
            import zlib
            trace = []
            crc = lambda t: zlib.crc32(t.detach().cpu().numpy().tobytes())
            for _ in range(n_steps):
                run.step(per_step)
                if drain:
                    torch.cuda.synchronize()
                    if os.environ.get("DIAG_TRACE") and run.learner is not None:
                        from distributed_drl_amd import _lib as L_
                        item = {"w": crc(run.learner.get_weights_flat()), "plan": [int(v) for v in run.last_plan]}
                        for nm, wh in (("g", L_.SAC1_GRAD), ("m", L_.SAC1_ADAM_M), ("t", L_.SAC1_TARGET)):
                            item[nm] = crc(run.learner.export(wh))
                        if os.environ.get("DIAG_TRACE") == "2":
                            item["_g"] = run.learner.export(L_.SAC1_GRAD).cpu().numpy()
                        for st in (0, 1):
                            item.update({"in%d_%s" % (st, k): crc(v) for k, v in run.learner.input_batch(st).items()})
                        item.update({"region%d" % o: crc(t) for o, t in run.regions.items()})
                        trace.append(item)
            torch.cuda.synchronize()
            run.check()
            out = {"trace": trace, "stats": {k: v for k, v in run.stats.items() if not k.startswith("s_")}}
            if run.learner is not None:
                out["w"] = run.learner.get_weights_flat().cpu().numpy()
                out["opt"] = run.learner.opt_steps()
            if run.rb is not None:
                out["counts"] = run.rb.get_counts()
                out["mt"] = run.rb.mt_state()[1]
                import zlib
                out["ring"] = {k: zlib.crc32(v.cpu().numpy().tobytes()) for k, v in run.rb.rings().items()}
            if run.roll is not None:
                out["pi"] = run.roll.actor.get_weights_flat().cpu().numpy()
            comm.barrier()
            return out

        a, a2, b = one(True), one(True), one(False)
        for st, (u, v) in enumerate(zip(a["trace"], a2["trace"])):
            diff = sorted(k for k in u if not k.startswith("_") and u[k] != v[k])
            extra = ""
            if diff and "_g" in u:
                ix = np.flatnonzero(u["_g"] != v["_g"])
                extra = " grad differs in %d elements: %s ... %s; max abs diff %.3g of max %.3g" % (ix.size, ix[:12].tolist(), ix[-4:].tolist(),
                        float(np.abs(u["_g"] - v["_g"]).max()), float(np.abs(u["_g"]).max()))
            assert not diff, "drained twice: step %d differs in %s (plan %s)%s" % (st, diff, u["plan"], extra)
        for tag, x in (("drained twice", a2), ("running ahead", b)):
            assert a["stats"] == x["stats"], (tag, a["stats"], x["stats"])
            for k in ("opt", "counts", "mt", "ring"):
                assert a.get(k) == x.get(k), (tag, k, a.get(k), x.get(k))
            for k in ("w", "pi"):
                if k in a:
                    bad = np.flatnonzero(a[k] != x[k])
                    assert bad.size == 0, "%s: %s differs in %d of %d elements, first at %d" % (tag, k, bad.size, a[k].size, bad[0])
        if roles.is_learner:
            assert a["opt"] == (n_steps * per_step,) * 2 and a["stats"]["pushes"] == 1 + (n_steps * per_step) // opt.push_freq
        q.put((rank, "ok"))
    except Exception:  # noqa
        import traceback
        q.put((rank, "FAIL: " + traceback.format_exc()))


def _gpu_worker_free(rank, world, port, q, num_learners, sizes=TOY, K=3):
    """FREE-RUNNING mode (PartitionedRun(free_steps=K); example/dsac.py:229-236: no gate): K vector steps per rollout rank and step,
    the blocks of step s + 1 drawn and sent behind the env steps of step s and received under the updates of step s, pushes adopted
    by the rollout ranks one step later off a communication stream.  Checked batch by batch against NumPy's own stream with the
    provenance the mode defines: update s trained on the batch the scheduled owner's ring handed out when it held the env steps
    of steps 0 .. s - 1 (step 0: the rings as constructed)."""
    try:
        n_envs, batch, cap, prefill, n_upd = sizes
        sys.path.insert(0, ROOT)
        os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                          DDRL_DIST_BACKEND="gloo")
        import distributed_drl_amd as d
        import torch.distributed as dist
        from distributed_drl_amd import _lib, comm, partition
        from distributed_drl_amd.agent import HyperParameters, Learner
        from distributed_drl_amd.workers import RolloutDevice
        r, w, _ = comm.init_from_env()
        torch.cuda.set_device(0)
        _lib.require_gpu()
        opt = HyperParameters()
        opt.num_envs, opt.batch_size, opt.seed, opt.start_steps, opt.max_ep_len, opt.push_freq = n_envs, batch, 5, -1, 50, 3
        B = opt.batch_size
        roles = partition.Roles(w, r, num_learners=num_learners)
        run = partition.PartitionedRun(opt, roles, lambda: _prefilled_shard(d, opt, r, cap, prefill), lambda rb: RolloutDevice(None, rb, opt, worker_index=r),
                                       lambda: Learner(opt, job="learner", index=0), seed=9, updates_per_graph=0, free_steps=K)
        n_pi = 8 * 400 + 400 + 400 * 300 + 300 + 2 * (300 * 2 + 2)
        sched = partition.Schedule(roles, seed=9)
        per_step = 2                                   # updates per learner and step: a push (every 3 updates) inside every other step
        n_steps = n_upd // per_step
        plans = [[sched.next() for _ in range(per_step)] for _ in range(n_steps + 1)]   # (+1: the blocks drawn ahead for a step that never runs)
        rs = np.random.RandomState(100 + r)
        handed, trained, adopted, pushed = [], [], [], []
        for s_ in range(n_steps):
            if s_ == 2:
                run.check()                            # a drain between steps (bench.py does one behind its timed region) changes nothing:
            run.step(per_step)                         # the blocks already posted for this step stay posted
            torch.cuda.synchronize()
            if run.rb is not None:
                # this ring's sampler has now drawn, in this order: (first call only) the blocks of step 0 for REMOTE learners from the ring
                # as constructed; behind the K env steps of this step, the blocks of step s + 1 for remote learners; then — a rank that
                # learns from its own shard (config 3's rank 0) — the local batches of step s, one per update, at update time
                after = min(cap, prefill + opt.num_envs * K * (s_ + 1))
                for kind, st, size in ((("remote", 0, prefill),) if s_ == 0 else ()) + (("remote", s_ + 1, after), ("local", s_, after)):
                    for li, l in enumerate(roles.learners):          # a shard serves learner 0's batches of a step first
                        if (l == r) != (kind == "local"):
                            continue
                        for u, p in enumerate(plans[st]):
                            if p[li][1] == r:
                                handed.append((st, u, l, _rows(run.rb, rs.randint(0, size, B))))
            if run.learner is not None:
                v = run.learner.input_batch(run.learner._dp_last_set if run.loop is None else (per_step * s_ + per_step - 1) & 1)
                trained.append(torch.cat([v[k].reshape(-1) for k in ("obs1", "obs2", "acts", "rews", "done")]).cpu().numpy().copy())
            if run.roll is not None:
                adopted.append(run.roll.actor.get_weights_flat().cpu().numpy().copy())
            pushed.append((run.stats["pushes"], run.bcast.buf[:n_pi].cpu().numpy().copy()))
        everything = [None] * w
        dist.all_gather_object(everything, handed)
        if roles.is_learner:
            want = {(st, u, l): rows for per_rank in everything for (st, u, l, rows) in per_rank}
            for s_ in range(n_steps):     # (the last update of every step is the one still lying in an input set)
                np.testing.assert_array_equal(trained[s_], want[(s_, per_step - 1, r)], err_msg="last update of step %d, learner %d" % (s_, r))
            assert run.learner.opt_steps() == (n_steps * per_step,) * 2
            assert run.stats["remote_batches"] + run.stats["local_batches"] == n_steps * per_step
        if run.rb is not None:
            served = sum(1 for st in range(n_steps + 1) for p in plans[st] for l, owner in p if owner == r and l != r) + \
                sum(1 for st in range(n_steps) for p in plans[st] for l, owner in p if owner == r and l == r)
            samples, steps, size = run.rb.get_counts()
            assert samples == served and steps == prefill + n_steps * K * opt.num_envs and size == min(cap, steps), (samples, served, steps)
        assert run.stats["pushes"] == 1 + (n_steps * per_step) // 3
        if run.roll is not None and not roles.is_learner:
            # a free-running rollout rank acts, during step s, on what the server held at the END of step s - 1 (the last push it saw
            # land before its env steps of step s were issued), never on a push of step s itself
            for s_ in range(1, n_steps):
                np.testing.assert_array_equal(adopted[s_], pushed[s_ - 1][1], err_msg="policy during step %d" % s_)
        if num_learners == 2:
            ws = [None] * w
            dist.all_gather_object(ws, run.learner.get_weights_flat().cpu().numpy() if roles.is_learner else None)
            np.testing.assert_array_equal(ws[0], ws[1])
        run.check()
        comm.barrier()
        q.put((rank, "ok"))
    except Exception:  # noqa
        import traceback
        q.put((rank, "FAIL: " + traceback.format_exc()))


def _spawn(world, num_learners, target=None, extra=(), timeout=300):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target or _gpu_worker, args=(r, world, port, q, num_learners) + tuple(extra)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=timeout) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(r, "ok") for r in range(world)], res


@pytest.mark.gpu
def test_config3_two_ranks_on_one_gpu_hip_ring_and_learner():
    """Config 3 end to end with the real HIP ring / sampler / learner loop on both ranks (gloo transport, both on cuda:0):
    every update trains on the batch the scheduled owner's ring hands out, counters and pushes line up on both ranks."""
    _spawn(2, None)


@pytest.mark.gpu
def test_config4_roles_three_ranks_on_one_gpu_two_learners_one_shard():
    """Config 4's roles at the smallest size (2 data-parallel learner ranks + 1 rollout rank with the shard): the owner
    serves both learners' blocks, the learners all-reduce their gradients and stay bit-identical."""
    _spawn(3, 2)


@pytest.mark.gpu
def test_free_running_mode_config3_roles_two_ranks_on_one_gpu():
    """PartitionedRun(free_steps=K) with config 3's roles (rank 0 learns AND rolls out, both ranks own a shard): local draws and
    prefetched remote blocks in one plan."""
    _spawn(2, None, target=_gpu_worker_free)


@pytest.mark.gpu
def test_free_running_mode_config4_roles_eight_ranks_on_one_gpu():
    """The free-running mode with config 4's roles — 8 ranks on the one GPU over gloo: 2 data-parallel learner ranks, 6 rollout ranks
    running K vector steps per step: every checked update trained on the batch NumPy's stream says the scheduled owner's ring handed
    out one step EARLIER (the blocks travel under the previous step's updates), each owner's sampler advanced once per batch it drew
    (one step ahead included), the rollout ranks act on the previous step's last push, the learners end identical."""
    _spawn(8, 2, target=_gpu_worker_free, timeout=600)


@pytest.mark.gpu
def test_config3_at_its_own_sizes_two_ranks_on_one_gpu():
    """BASELINE config 3 as stated — the 10^6-transition replay sharded over 2 ranks (500 000 each, wrapping during the run), 4096 envs per
    rank, batch 256, learner on rank 0 — batch by batch against NumPy's stream, like the toy run above."""
    _spawn(2, None, extra=(CONFIG3,), timeout=600)


@pytest.mark.gpu
def test_config4_at_its_own_sizes_eight_ranks_on_one_gpu():
    """BASELINE config 4 as it is stated — 8 ranks: 2 data-parallel learner ranks + 6 rollout ranks x 8192 envs, batch 256, 10^6
    transitions over 6 shards (algos/sac1/sac_ray.py:137-141,246,316-324; example/dsac.py:229-233) — on the ranks of the one GPU over
    gloo: every update's TWO batches (one per learner) are bit for bit what NumPy's own stream says the scheduled owner's ring hands out
    (the rings wrap during the run), every batch is remote (dedicated learners own no shard), each owner's sampler advanced once per
    batch it served, pushes reach all six rollout ranks' actors, and the two learners hold identical parameters at the end."""
    _spawn(8, 2, extra=(CONFIG4,), timeout=600)


@pytest.mark.gpu
@pytest.mark.parametrize("world,num_learners", [(2, None), (3, 2)])
def test_host_running_steps_ahead_of_the_device_changes_nothing(world, num_learners):
    """No per-step synchronisation, graph-captured learner loop with a feed attached, several updates per step: the end
    state equals the drained run's (staged plans / blocks are never rewritten under a step in flight)."""
    _spawn(world, num_learners, target=_gpu_worker_async)
