import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import distributed_drl_amd as d
rb = d.ReplayBufferSAC1(8, 2, 10**6, seed=0)
m = 10**6
rb.store_batch(torch.randn(m, 8).cuda(), torch.rand(m, 2).cuda(), torch.randn(m).cuda(), torch.randn(m, 8).cuda(), torch.zeros(m).cuda())
B, K = 256, 1024
flat = torch.empty(K * B * 20 + 64, device="cuda")
for _ in range(3): rb.sample_many(B, K, flat)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): rb.sample_many(B, K, flat)
torch.cuda.synchronize(); print("sample_many(256 x 1024): %.3f ms" % ((time.perf_counter() - t0) / 20 * 1e3))
