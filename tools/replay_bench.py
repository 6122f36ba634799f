#!/usr/bin/env python3
"""Replay store / sample micro-benchmark at the SAC1 shape (obs 8, act 2, capacity 1M):
algorithmic bytes per SURVEY §8(d): store 160 B/transition, sample 41 984 B per batch of 256."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import distributed_drl_amd as d
from distributed_drl_amd import _lib

_lib.require_gpu()
rb = d.ReplayBufferSAC1(8, 2, 10 ** 6, seed=0)
res = {}
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for n in (4096, 65536, 1000000):
    o, o2 = torch.randn(n, 8, device="cuda"), torch.randn(n, 8, device="cuda")
    a, r, dn = torch.rand(n, 2, device="cuda"), torch.randn(n, device="cuda"), torch.zeros(n, device="cuda")
    for _ in range(3):
        rb.store_batch(o, a, r, o2, dn)
    torch.cuda.synchronize()
    reps = 50
    e0.record()
    for _ in range(reps):
        rb.store_batch(o, a, r, o2, dn)
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / reps * 1e-3
    res["store_n%d" % n] = {"us": t * 1e6, "GBps": n * 160 / t / 1e9, "transitions_per_s": n / t}
for B in (256, 4096):
    for _ in range(3):
        rb.sample_batch_device(B)
    torch.cuda.synchronize()
    reps = 200
    e0.record()
    for _ in range(reps):
        rb.sample_batch_device(B)
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / reps * 1e-3
    res["sample_B%d" % B] = {"us": t * 1e6, "GBps": B * 164 / t / 1e9, "batches_per_s": 1 / t}
print(json.dumps(res))
