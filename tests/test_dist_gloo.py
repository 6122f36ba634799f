"""N > 1 path on CPU: world_size-2 gloo process group exercising comm.py (parameter broadcast, shard choice,
gradient mean) and partition.py's block transport (one point-to-point message per owner, learner and step)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    try:
        sys.path.insert(0, ROOT)
        os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                          MASTER_PORT=str(port))
        from distributed_drl_amd import comm
        from oracle.replay_oracle import ReplayBufferOracle
        r, w, _ = comm.init_from_env("gloo")
        assert (r, w) == (rank, world) and comm.rank() == rank and comm.world_size() == world
        dev = torch.device("cpu")

        # ---- ps.push/pull as one broadcast of the flat vector -------------------------------
        n = 375106
        pb = comm.ParamBroadcast(n, dev, src=0)
        pb.register_span(("main/pi/a", "main/pi/b"), 0, 125104)
        flat = torch.arange(n, dtype=torch.float32) * 0.5 if rank == 0 else None
        for push in range(2):
            if rank == 0:
                flat = flat + 1.0
            got = pb.sync(flat)
            want = torch.arange(n, dtype=torch.float32) * 0.5 + (push + 1)
            assert torch.equal(got, want) and pb.version == push + 1
        off, cnt = pb.span(("main/pi/a", "main/pi/b"))
        assert torch.equal(pb.pull_flat(off, cnt), want[:125104])

        # ---- replay shards: local store, shard choice on a shared stream, ONE P2P block per (owner, learner, step) ----
        from distributed_drl_amd import partition
        B, n_upd = 16, 11
        shard = ReplayBufferOracle(8, 2, 64, seed=100 + rank)   # test double for the device ring
        rs = np.random.RandomState(rank)
        for i in range(40):
            shard.store(rs.randn(8), rs.randn(2), float(rank * 1000 + i), rs.randn(8), i % 5 == 0)
        roles = partition.Roles(world, rank)                    # config 3: learner on rank 0, a shard on both ranks
        sched = partition.Schedule(roles, seed=7)
        nf = partition.batch_floats(8, 2, B)
        served = 0
        for step in range(2):
            plans = [sched.next() for _ in range(n_upd)]
            owners = [p[0][1] for p in plans]
            if step == 0:
                np.random.seed(7)   # the reference's call: np.random.choice(num_shards, 1)[0]
                assert owners == [int(np.random.choice(world, 1)[0]) for _ in range(n_upd)]
            k = owners.count(1)
            if rank == 1:           # owner: the k batches it owes the learner, as one block [obs1 | obs2 | acts | rews | done]
                drawn = [shard.sample_batch(B) for _ in range(k)]
                blk = torch.from_numpy(np.concatenate([np.concatenate([b[key].reshape(-1) for b in drawn])
                                                       for key in ("obs1", "obs2", "acts", "rews", "done")]).astype(np.float32))
                assert blk.numel() == k * nf
                partition._send(blk, 0).wait()
                served += k
                assert shard.sample_times == served
            else:                   # learner: one receive; batch i of the block = rows [i B, (i + 1) B) of every array
                got = partition._Recv(torch.empty(k * nf), 1).wait()
                rews = got[k * B * 18: k * B * 19].view(k, B)
                assert ((rews >= 1000) & (rews < 1040)).all()   # a batch never mixes shards (sac_ray.py:137-141)
                v = partition.batch_views(got[: nf], 8, 2, B) if k == 1 else None
                assert v is None or v["obs1"].shape == (B, 8)
                assert shard.sample_times == 0                  # only the owner's stream advanced for the remote batches

        # ---- learner gradient all-reduce (mean) ------------------------------------------------
        g = torch.full((1000,), float(rank + 1))
        comm.allreduce_mean_(g)
        assert torch.allclose(g, torch.full((1000,), (1 + world) / 2.0 if world == 2 else g[0].item()))
        assert comm.allreduce_max(float(rank)) == world - 1
        assert comm.allreduce_sum(1.0) == world
        comm.barrier()
        q.put((rank, "ok"))
    except Exception as e:  # noqa
        import traceback
        q.put((rank, "FAIL: " + traceback.format_exc()))


def test_two_rank_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(30)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res


def test_single_process_fallbacks():
    sys.path.insert(0, ROOT)
    from distributed_drl_amd import comm
    assert comm.world_size() == 1 and comm.rank() == 0
    g = torch.ones(4)
    assert comm.allreduce_mean_(g) is g and comm.allreduce_max(3.0) == 3.0
    pb = comm.ParamBroadcast(10, torch.device("cpu"))
    assert torch.equal(pb.sync(torch.arange(10.0)), torch.arange(10.0))
