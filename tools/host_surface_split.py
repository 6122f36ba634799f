"""Where the host-buffer loop train(sample_batch()) spends its time: the train side alone (a fixed host batch), the sample side alone (prefetch on the
buffer's own stream), both, and a cProfile of the loop.  python tools/host_surface_split.py   (profiles/r06_host_surface.txt)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import distributed_drl_amd as d
from distributed_drl_amd.agent import HyperParameters, Learner
opt = HyperParameters(); opt.batch_size = 256
rb = d.ReplayBufferSAC1(8, 2, 10 ** 6, seed=0)
rs = np.random.RandomState(0)
n = 200000
rb.store_batch(*(torch.from_numpy(x).cuda() for x in (rs.randn(n, 8).astype(np.float32), rs.uniform(-1, 1, (n, 2)).astype(np.float32), rs.randn(n).astype(np.float32), rs.randn(n, 8).astype(np.float32), np.zeros(n, np.float32))))
L = Learner(opt)
def timed(f, k=3000, warm=300):
    for _ in range(warm): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / k * 1e6
rb.prefetch(256, own_stream=True)
b = {k: v.copy() for k, v in rb.sample_batch(256).items()}
print("train(fixed host batch) only: %.1f us" % timed(lambda: L.train(b)))
print("sample_batch (prefetch own stream) only: %.1f us" % timed(lambda: rb.sample_batch(256)))
print("both: %.1f us" % timed(lambda: L.train(rb.sample_batch(256))))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(2000): L.train(rb.sample_batch(256))
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
