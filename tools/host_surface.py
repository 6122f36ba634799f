import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import distributed_drl_amd as d
from distributed_drl_amd.agent import HyperParameters, Learner
opt = HyperParameters(); B = 256; opt.batch_size = B
rbh = d.ReplayBufferSAC1(8, 2, 100000, seed=5)
rs = np.random.RandomState(0)
oo, aa = rs.randn(8), rs.uniform(-1, 1, 2).astype(np.float32)
for _ in range(200): rbh.store(oo, aa, 0.5, oo, False)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(2000): rbh.store(oo, aa, 0.5, oo, False)
torch.cuda.synchronize(); t_store = (time.perf_counter() - t0) / 2000
m = 100000
rbh.store_batch(torch.randn(m, 8).cuda(), torch.rand(m, 2).cuda(), torch.randn(m).cuda(), torch.randn(m, 8).cuda(), torch.zeros(m).cuda())
for _ in range(20): bh = rbh.sample_batch(B)
t0 = time.perf_counter()
for _ in range(300): bh = rbh.sample_batch(B)
t_samp = (time.perf_counter() - t0) / 300
lh = Learner(opt, job="learner", index=77)
for _ in range(5): lh.train(bh)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(300): lh.train(rbh.sample_batch(B))
torch.cuda.synchronize(); t_iter = (time.perf_counter() - t0) / 300
print("store %.1f us  sample_batch %.1f us  sample+train %.1f us" % (t_store * 1e6, t_samp * 1e6, t_iter * 1e6))
