#!/usr/bin/env python3
"""Dev helper for counter passes: N eager SAC1 updates in the learner loop (every dispatch 'in situ': caches as the update
sequence leaves them), nothing else.  rocprofv3 --pmc ... -- python3 tools/insitu.py 40"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import distributed_drl_amd as d
from distributed_drl_amd import _lib
from distributed_drl_amd.agent import HyperParameters
from distributed_drl_amd.workers import TrainDevice
n_upd = int(sys.argv[1]) if len(sys.argv) > 1 else 40
_lib.require_gpu()
opt = HyperParameters()
rb = d.ReplayBufferSAC1(8, 2, 1 << 17, seed=0)
n = 1 << 17
rs = np.random.RandomState(0)
rb.store_batch(*(torch.from_numpy(x).cuda() for x in (rs.randn(n, 8).astype(np.float32), rs.uniform(-1, 1, (n, 2)).astype(np.float32),
                                                      rs.randn(n).astype(np.float32), rs.randn(n, 8).astype(np.float32),
                                                      (rs.rand(n) < 0.01).astype(np.float32))))
td = TrainDevice(None, rb, opt, updates_per_graph=0)
td.run(n_upd)
torch.cuda.synchronize()
print("insitu: %d eager updates done" % n_upd)
