#!/usr/bin/env python3
"""HBM-bound gather stress (BASELINE config 5 row shape: flat 84x84x4 observations, 225 804 B per
transition, batch 512).  The ring is sized well beyond the 256 MiB Infinity Cache so that the rows
really come from HBM.  Reports algorithmic GB/s = B*(2*T + 4) bytes / time for
(a) the stand-alone gather with caller indices and (b) the full sample_batch (MT19937 + gather)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import distributed_drl_amd as d
from distributed_drl_amd import _lib

_lib.require_gpu()
obs_dim, B = 84 * 84 * 4, 512
cap = int(sys.argv[1]) if len(sys.argv) > 1 else 65536      # 65536 x 225.8 KB = 14.8 GB


class Opt:
    pass


Opt.obs_dim, Opt.buffer_size, Opt.batch_size, Opt.save_dir = obs_dim, cap, B, "."
rb = d.ReplayBufferDQN(Opt, 0, seed=0)
# fill on the device (content irrelevant for timing; rows must exist)
chunk = 2048
g = torch.Generator(device="cuda").manual_seed(0)
for s in range(0, cap, chunk):
    n = min(chunk, cap - s)
    o = torch.randint(0, 256, (n, obs_dim), device="cuda", generator=g).float()
    rb.store_batch(o, torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda"), o, torch.zeros(n, device="cuda"))
torch.cuda.synchronize()
T = 4 * (2 * obs_dim + 1 + 2)
bytes_per_batch = B * (2 * T + 4)
res = {}
# (a) gather with caller-supplied indices, a fresh index set per call
idx = [torch.randint(0, cap, (B,), device="cuda", generator=g) for _ in range(64)]
for warm in range(3):
    rb.gather_device(idx[warm], fresh=False)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(64):
    rb.gather_device(idx[i], fresh=False)
e1.record()
torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 64 * 1e-3
res["gather_us"] = t * 1e6
res["gather_GBps"] = bytes_per_batch / t / 1e9
# (b) full sample_batch
for warm in range(3):
    rb.sample_batch_device(B)
torch.cuda.synchronize()
e0.record()
for i in range(64):
    rb.sample_batch_device(B)
e1.record()
torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 64 * 1e-3
res["sample_batch_us"] = t * 1e6
res["sample_batch_GBps"] = bytes_per_batch / t / 1e9
res.update(capacity=cap, ring_GB=cap * T / 1e9, batch=B, bytes_per_batch=bytes_per_batch,
           hbm_peak_GBps=8000, frac_of_peak=res["gather_GBps"] / 8000)
print(json.dumps(res))
