"""Host cost of one data-parallel learner iteration (config 4) without the all-reduce: is the eager loop host-bound?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import distributed_drl_amd as d
from distributed_drl_amd.agent import HyperParameters, Learner

opt = HyperParameters(); opt.batch_size = 256; opt.seed = 0
rb = d.ReplayBufferSAC1(8, 2, 100000, seed=1)
rs = np.random.RandomState(0); n = 100000
rb.store_batch(*(torch.from_numpy(x).cuda() for x in (rs.randn(n, 8).astype(np.float32), rs.uniform(-1, 1, (n, 2)).astype(np.float32),
                                                       rs.randn(n).astype(np.float32), rs.randn(n, 8).astype(np.float32), np.zeros(n, np.float32))))
L = Learner(opt, job="learner", index=0)
grads, apply, g = L.dp_stepper(rb)


def it(last=False):
    grads(last)
    apply()


for _ in range(50):
    it()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(1000):
    it()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("DP learner iteration without the all-reduce: host %.1f us/update to issue, %.1f us/update to complete" % ((t1 - t0) * 1e3, (t2 - t0) * 1e3))
