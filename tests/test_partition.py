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


def _gpu_worker(rank, world, port, q):
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
        opt.num_envs, opt.batch_size, opt.seed, opt.start_steps, opt.max_ep_len, opt.push_freq = 64, 32, 5, -1, 50, 6
        roles = partition.Roles(w, r)

        def make_replay():   # every row of shard s carries reward 1000 s + i: a batch shows which ring it came from
            rb = d.ReplayBufferSAC1(opt.obs_dim, opt.act_dim, 4096, seed=100 + r)
            rs = np.random.RandomState(r)
            n = 500
            rb.store_batch(*(torch.from_numpy(x).cuda() for x in (
                rs.randn(n, 8).astype(np.float32), rs.uniform(-1, 1, (n, 2)).astype(np.float32),
                (1000.0 * r + np.arange(n)).astype(np.float32), rs.randn(n, 8).astype(np.float32), np.zeros(n, np.float32))))
            return rb
        run = partition.PartitionedRun(opt, roles, make_replay, lambda rb: RolloutDevice(None, rb, opt, worker_index=r),
                                       lambda: Learner(opt, job="learner", index=r), seed=9)
        # the learner's initial weights reached the other rank's actor
        pi0 = run.bcast.buf[: run.roll.actor.n_params].clone()
        assert torch.equal(run.roll.actor.get_weights_flat(), pi0)
        sched = partition.Schedule(roles, seed=9)
        n_upd = 14
        owners = [sched.next()[0][1] for _ in range(n_upd)]
        # record what every ring hands out (in order) and what the learner trains on (in order)
        drawn, trained = [], []
        orig = partition.sample_packed

        def recording(rb, B, flat):
            v = orig(rb, B, flat)
            drawn.append(flat.detach().cpu().numpy().copy())
            return v
        partition.sample_packed = recording
        for u in range(n_upd):   # one update per step() call so that the batch of every update can be inspected
            run.step(1)
            if roles.is_learner:
                trained.append(torch.cat([run.last_batch[k].reshape(-1) for k in ("obs1", "obs2", "acts", "rews", "done")]).cpu().numpy().copy())
        torch.cuda.synchronize()
        import torch.distributed as dist
        all_drawn = [None, None]
        dist.all_gather_object(all_drawn, drawn)
        if roles.is_learner:
            # update u trained on exactly the next batch drawn from the ring of the scheduled owner (bit for bit; never a mix)
            nxt = [0, 0]
            for u, own in enumerate(owners):
                np.testing.assert_array_equal(trained[u], all_drawn[own][nxt[own]], err_msg="update %d from shard %d" % (u, own))
                nxt[own] += 1
            assert nxt == [len(all_drawn[0]), len(all_drawn[1])]
            assert run.learner.opt_steps() == (n_upd, n_upd)
            assert run.stats["local_batches"] == owners.count(0) and run.stats["remote_batches"] == owners.count(1)
        else:
            assert run.stats["sent_batches"] == owners.count(1) == len(drawn)
        # owner-side sampler streams advanced exactly once per batch they served; both rings took their local stores
        samples, steps, size = run.rb.get_counts()
        assert samples == owners.count(r) and steps == 500 + n_upd * opt.num_envs
        # two pushes (updates 6 and 12) + the initial one: the remote actor runs the learner's pushed policy
        assert run.stats["pushes"] == 3
        flat = run.bcast.buf.clone()
        if roles.is_learner:
            assert not torch.equal(flat[: pi0.numel()], pi0)
        assert torch.equal(run.roll.actor.get_weights_flat(), flat[: pi0.numel()])
        comm.barrier()
        q.put((rank, "ok"))
    except Exception:  # noqa
        import traceback
        q.put((rank, "FAIL: " + traceback.format_exc()))


@pytest.mark.gpu
def test_config3_two_ranks_on_one_gpu_hip_ring_and_learner():
    """Config 3 end to end with the real HIP ring / sampler / learner on both ranks (gloo transport, both on cuda:0): the
    learner's batches come from the scheduled owner's ring, counters and pushes line up on both ranks."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gpu_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res
