#!/usr/bin/env python3
"""Dev helper: per-stage launch times of one SAC1 update (back-to-back launches between two HIP
events) and the graph-loop rate.  Usage: python tools/stage_times.py [reps]"""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import distributed_drl_amd as d
from distributed_drl_amd import _lib
from distributed_drl_amd.agent import HyperParameters
from distributed_drl_amd.workers import TrainDevice

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
nograph = len(sys.argv) > 2 and sys.argv[2] == 'nograph'  # PMC collection crashes on graph launches
_lib.require_gpu()
opt = HyperParameters()
rb = d.ReplayBufferSAC1(8, 2, 10 ** 6, seed=0)
n = 1 << 17
rs = np.random.RandomState(0)
for _ in range(8):
    rb.store_batch(*(torch.from_numpy(x).cuda() for x in (rs.randn(n, 8).astype(np.float32), rs.uniform(-1, 1, (n, 2)).astype(np.float32),
                                                          rs.randn(n).astype(np.float32), rs.randn(n, 8).astype(np.float32),
                                                          (rs.rand(n) < 0.01).astype(np.float32))))
td = TrainDevice(None, rb, opt, updates_per_graph=0 if nograph else 32)
td.run(8 if nograph else 64)
torch.cuda.synchronize()
lib = _lib.load()
names = ["-", "-", "k_dfwd<0>", "-", "-", "k_dfwd<1>", "-", "k_dg bq", "k_dg mid", "k_dg pi", "-"]   # direct path: 5 launches per update
ms = ctypes.c_float()
tot = 0.0
for st in range(1, 11):
    _lib.check(lib.ddrl_sac1_stage_time(td.agent._h, st, reps, ctypes.byref(ms), _lib.stream_ptr()))
    print("stage %2d %-10s %8.2f us" % (st, names[st], ms.value * 1e3))
    tot += ms.value * 1e3
print("sum of stages 1-10: %.1f us" % tot)
torch.cuda.synchronize()
if nograph:
    sys.exit(0)
t0 = time.perf_counter()
td.run(2048)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("graph loop: %.1f us/update  (%.0f updates/s)" % (dt / 2048 * 1e6, 2048 / dt))
