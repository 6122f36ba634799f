"""N > 1 path on CPU: world_size-2 gloo process group exercising comm.py (parameter broadcast,
shard choice, batch fetch from the shard owner, gradient mean)."""
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

        # ---- replay shards: local store, shard choice on a shared stream, owner gathers ------
        shard = ReplayBufferOracle(8, 2, 64, seed=100 + rank)   # test double for the device ring
        rs = np.random.RandomState(rank)
        for i in range(40):
            shard.store(rs.randn(8), rs.randn(2), float(rank * 1000 + i), rs.randn(8), i % 5 == 0)
        picker = comm.ShardPicker(world, seed=7)
        like = dict(obs1=torch.empty(16, 8), obs2=torch.empty(16, 8), acts=torch.empty(16, 2),
                    rews=torch.empty(16), done=torch.empty(16))
        owners = []
        for it in range(6):
            owner = picker.next()
            owners.append(owner)
            b = comm.fetch_batch(lambda: {k: torch.from_numpy(v) for k, v in shard.sample_batch(16).items()},
                                 owner, like)
            # a batch never mixes shards (sac_ray.py:137-141): rewards carry the owner's tag
            assert ((b["rews"] >= owner * 1000) & (b["rews"] < owner * 1000 + 40)).all()
            assert b["obs1"].shape == (16, 8) and b["acts"].shape == (16, 2)
        np.random.seed(7)   # the reference's call: np.random.choice(num_shards, 1)[0]
        assert owners == [int(np.random.choice(world, 1)[0]) for _ in range(6)]
        assert shard.sample_times == owners.count(rank)   # only the owner's stream advanced

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
    b = comm.fetch_batch(lambda: dict(obs1=torch.ones(2, 8), obs2=torch.zeros(2, 8), acts=torch.ones(2, 2),
                                      rews=torch.arange(2.0), done=torch.zeros(2)),
                         0, dict(obs1=torch.empty(2, 8), obs2=torch.empty(2, 8), acts=torch.empty(2, 2),
                                 rews=torch.empty(2), done=torch.empty(2)))
    assert torch.equal(b["rews"], torch.arange(2.0))
