"""Soak: two identical learner loops of N updates each (graph-captured, sampler riding along) must end bit-identical and finite —
the hand-offs inside the launches (last-arriving row tile, sampler workgroup) decide who does the work, never the result."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import distributed_drl_amd as d
from distributed_drl_amd import _lib
from distributed_drl_amd.agent import HyperParameters
from distributed_drl_amd.workers import TrainDevice

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
per_graph = int(sys.argv[3]) if len(sys.argv) > 3 else 50
outs = []
for rep in range(2):
    opt = HyperParameters(); opt.batch_size, opt.seed, opt.push_freq = B, 1, 10 ** 9
    rb = d.ReplayBufferSAC1(8, 2, 200000, seed=5)
    rs = np.random.RandomState(0); m = 200000
    rb.store_batch(*(torch.from_numpy(x).cuda() for x in (rs.randn(m, 8).astype(np.float32), rs.uniform(-1, 1, (m, 2)).astype(np.float32),
                                                           rs.randn(m).astype(np.float32), rs.randn(m, 8).astype(np.float32),
                                                           (rs.rand(m) < 0.01).astype(np.float32))))
    td = TrainDevice(None, rb, opt, learner_index=0, updates_per_graph=per_graph)
    td.run(n)
    torch.cuda.synchronize()
    outs.append([td.agent.export(w).clone() for w in (_lib.SAC1_MAIN, _lib.SAC1_TARGET, _lib.SAC1_ADAM_M, _lib.SAC1_ADAM_V)] + [td.agent.opt_steps()])
for a, b in zip(outs[0][:4], outs[1][:4]):
    assert torch.isfinite(a).all(), "non-finite parameters"
    assert torch.equal(a, b), "two identical runs differ"
assert outs[0][4] == outs[1][4] == (n, n)
print("soak ok (batch %d, %d per graph): %d updates twice, bit-identical, finite; |main|max = %.3f" % (B, per_graph, n, outs[0][0].abs().max().item()))
