"""Determinism hunt for the partitioned run, single process: (a) learner loop with the rollout stepping between steps, no feed;
(b) the same plus a feed plan over a second ring's blocks (the two-rank data flow without the transport)."""
import os, sys, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import distributed_drl_amd as d
from distributed_drl_amd import partition
from distributed_drl_amd.agent import HyperParameters, Learner
from distributed_drl_amd.workers import RolloutDevice

mode = sys.argv[1] if len(sys.argv) > 1 else "a"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
per_graph = int(sys.argv[3]) if len(sys.argv) > 3 else 0
with_roll = int(sys.argv[4]) if len(sys.argv) > 4 else 1
opt = HyperParameters()
opt.num_envs, opt.batch_size, opt.seed, opt.start_steps, opt.max_ep_len, opt.push_freq = 64, 32, 5, -1, 50, 1000
B, nf = 32, 32 * 20

def shard(r):
    rb = d.ReplayBufferSAC1(8, 2, 4096, seed=100 + r)
    rs = np.random.RandomState(r); n = 500
    rb.store_batch(*(torch.from_numpy(x).cuda() for x in (
        rs.randn(n, 8).astype(np.float32), rs.uniform(-1, 1, (n, 2)).astype(np.float32),
        (1000.0 * r + np.arange(n)).astype(np.float32), rs.randn(n, 8).astype(np.float32), np.zeros(n, np.float32))))
    return rb

def one():
    rb = shard(0)
    learner = Learner(opt, job="learner", index=0)
    roll = RolloutDevice(None, rb, opt, worker_index=0) if with_roll else None
    loop = partition._Loop(learner, rb, per_graph)
    other = shard(1) if mode == "b" else None
    rs = np.random.RandomState(9)
    ws = []
    for step in range(9):
        if roll is not None:
            roll.step()
        if other is not None:
            owners = rs.randint(0, 2, 5)
            k = int((owners == 1).sum())
            regions = []
            if k:
                blk = other.sample_many(B, k, torch.empty(max(k, 64) * nf, dtype=torch.float32, device="cuda"))
                regions = [(blk, k)]
            plan, nxt = [], 0
            for o in owners:
                if o == 0: plan.append(-1)
                else: plan.append(nxt); nxt += 1
            rb.set_feed(torch.tensor(plan, dtype=torch.int32, device="cuda"), B, regions)
        loop.run(5)
        torch.cuda.synchronize()
        ws.append(zlib.crc32(learner.get_weights_flat().cpu().numpy().tobytes()))
    return ws

poison = int(sys.argv[5]) if len(sys.argv) > 5 else 0
runs = []
for rep in range(reps):
    if poison:   # scramble what freed device memory holds and the order the allocator hands it out
        g = torch.Generator(device="cuda").manual_seed(rep)
        junk = [torch.randint(-2 ** 31, 2 ** 31 - 1, (sz,), dtype=torch.int32, device="cuda", generator=g) for sz in (1 << 26, 1 << 22, 1 << 20, 12345)]
        torch.cuda.synchronize()
        del junk
        torch.cuda.empty_cache()
        keep = [torch.empty((rep + 1) * 100000, device="cuda")]
    runs.append(one())
bad = [i for i, r in enumerate(runs) if r != runs[0]]
print("part_det mode %s per_graph %d roll %d: %s" % (mode, per_graph, with_roll, "all %d runs identical" % reps if not bad else
      "runs %s differ from run 0, first at step %s" % (bad, [next(j for j in range(9) if runs[i][j] != runs[0][j]) for i in bad])))
