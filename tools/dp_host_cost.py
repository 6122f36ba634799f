"""Cost of one data-parallel learner iteration (config 4): eager (gradients -> all-reduce -> apply, one launch sequence per update)
vs the captured form (graphs of k updates with the RCCL all-reduce as a graph node, partition.PartitionedRun._capture_dp).
Runs a world-size-1 RCCL group on the one GPU: the collective is real (an RCCL kernel per update), its wire time is not."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", DDRL_DIST_FORCE="1")
os.environ.setdefault("MASTER_PORT", "29531")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import numpy as np
import torch
import torch.distributed as dist
import distributed_drl_amd as d
from distributed_drl_amd import comm
from distributed_drl_amd.agent import HyperParameters, Learner

comm.init_from_env()
grp = dist.new_group(ranks=[0])
opt = HyperParameters(); opt.batch_size = 256; opt.seed = 0
rb = d.ReplayBufferSAC1(8, 2, 100000, seed=1)
rs = np.random.RandomState(0); n = 100000
rb.store_batch(*(torch.from_numpy(x).cuda() for x in (rs.randn(n, 8).astype(np.float32), rs.uniform(-1, 1, (n, 2)).astype(np.float32),
                                                       rs.randn(n).astype(np.float32), rs.randn(n, 8).astype(np.float32), np.zeros(n, np.float32))))
L = Learner(opt, job="learner", index=0)
grads, apply, g = L.dp_stepper(rb)


def it(last=False, reduce=True):
    grads(last)
    if reduce:
        comm.allreduce_mean_(g, group=grp)
    apply()


def timed(fn, n_upd, label):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    fn()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("%-64s host %6.1f us/update to issue, %6.1f us/update to complete" % (label, (t1 - t0) / n_upd * 1e6, (t2 - t0) / n_upd * 1e6))


for _ in range(50):
    it()
it(last=True)
timed(lambda: [it(reduce=False) for _ in range(999)] + [it(last=True, reduce=False)], 1000, "eager, no all-reduce")
timed(lambda: [it() for _ in range(999)] + [it(last=True)], 1000, "eager + RCCL all-reduce (AVG) per update")
K = 16
for reduce, label in ((False, "graphs of %d updates, NO all-reduce (the step's own device time)" % K), (True, "graphs of %d updates, RCCL all-reduce as a graph node" % K)):
    grads.graph_sync()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for i in range(K):
            it(last=(i == K - 1), reduce=reduce)
        grads.graph_sync()
    graph.replay(); torch.cuda.synchronize()
    timed(lambda: [graph.replay() for _ in range(64)], 64 * K, label)
dist.destroy_process_group()
