#!/usr/bin/env python3
"""Dev probe (lever i): is the update's time set by tile counts that do not divide into 256 CUs?
The five launches of the direct-operand update at batch 256 over a sweep of hidden sizes: per launch the number of 32x32 tiles,
the warm back-to-back launch time (ddrl_sac1_stage_time) and the time per update in the graph loop.  A quantisation tail would
show as a STEP where a launch's tile count crosses 256 (a second workgroup on some CUs) and a flat stretch up to 512."""
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
_lib.require_gpu()
lib = _lib.load()
rb = d.ReplayBufferSAC1(8, 2, 1 << 18, seed=0)
n = 1 << 17
rs = np.random.RandomState(0)
for _ in range(2):
    rb.store_batch(*(torch.from_numpy(x).cuda() for x in (rs.randn(n, 8).astype(np.float32), rs.uniform(-1, 1, (n, 2)).astype(np.float32),
                                                          rs.randn(n).astype(np.float32), rs.randn(n, 8).astype(np.float32),
                                                          (rs.rand(n) < 0.01).astype(np.float32))))
B = 256
print("batch %d; tiles = 32x32 output tiles per launch; us = warm back-to-back launches; loop = graph loop per update" % B)
print("%5s %5s | %-13s %-13s %-13s %-13s %-13s | %7s %7s | %s" % ("h1", "h2", "k_dfwd<0>", "k_dfwd<1>", "k_dg bq", "k_dg mid", "k_dg pi", "sum", "loop", "MFLOP/update"))
shapes = [(h1, 300) for h1 in (256, 288, 320, 352, 384, 400, 416, 448, 480, 512)] + [(400, h2) for h2 in (192, 224, 256, 288, 320, 352, 384)]
for h1, h2 in shapes:
    opt = HyperParameters()
    opt.hidden_sizes = (h1, h2)
    opt.batch_size = B
    td = TrainDevice(None, rb, opt, updates_per_graph=32)
    td.run(64)
    torch.cuda.synchronize()
    rt, c1, c2 = B // 32, -(-h1 // 32), -(-h2 // 32)
    c1b = -(-(h1 + 1) // 32)                       # wgrad rows incl. the bias row
    tiles = {2: 3 * rt * c2, 5: 5 * rt * c2 + 1, 7: 3 * rt * c1, 8: rt * c1 + rt * c2 + 2 * c1b * c2 + 2 * -(-(h2 + 1) // 32),
             9: c1b * c2 + 2 * -(-(h2 + 1) // 32) + 3 * c1 + 1}
    ms = ctypes.c_float()
    cells, tot = [], 0.0
    for st in (2, 5, 7, 8, 9):
        _lib.check(lib.ddrl_sac1_stage_time(td.agent._h, st, reps, ctypes.byref(ms), _lib.stream_ptr()))
        cells.append("%4d %6.2f  " % (tiles[st], ms.value * 1e3))
        tot += ms.value * 1e3
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    td.run(4096)
    torch.cuda.synchronize()
    loop = (time.perf_counter() - t0) / 4096 * 1e6
    macs = (3 * (9 * h1 + h1 * h2 + h2 * 4) + 5 * (11 * h1 + h1 * h2 + h2)) + (2 * (9 * h1 + h1 * h2 + 4 * h2) - 9 * h1 + (h1 * h2 + h2) ) + 2 * 2 * (11 * h1 + h1 * h2 + h2)
    print("%5d %5d | %s| %7.2f %7.2f | %.0f" % (h1, h2, "".join(cells), tot, loop, 2 * B * macs / 1e6), flush=True)
    del td
