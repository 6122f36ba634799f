"""Timeline of FreeRunningLoop's phases (ms since a base event) for a few segments."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import distributed_drl_amd as d
from distributed_drl_amd.agent import HyperParameters
from distributed_drl_amd.workers import RolloutDevice, TrainDevice, FreeRunningLoop
opt = HyperParameters(); opt.num_envs, opt.batch_size, opt.start_steps, opt.max_ep_len, opt.seed = 4096, 256, -1, 1000, 0
rb = d.ReplayBufferSAC1(8, 2, 1000000, seed=0)
n = 1000000
g = torch.Generator(device="cuda").manual_seed(1)
rb.store_batch(torch.randn(n, 8, device="cuda", generator=g), torch.rand(n, 2, device="cuda", generator=g), torch.randn(n, device="cuda", generator=g),
               torch.randn(n, 8, device="cuda", generator=g), torch.zeros(n, device="cuda"))
tr = TrainDevice(None, rb, opt, updates_per_graph=50)
ps = d.ParameterServer(*tr.agent.get_weights()); tr.ps = ps
ro = RolloutDevice(ps, rb, opt)
tr.run(100); ro.step(5); torch.cuda.synchronize()
K = int(os.environ.get("K", "100"))
loop = FreeRunningLoop(ro, tr, opt, steps_per_segment=K, updates_per_segment=100, timing=True)
loop.run(3); loop.drain()
base = torch.cuda.Event(enable_timing=True); base.record(); torch.cuda.synchronize()
n0 = len(loop.marks)
t0 = time.perf_counter()
loop.run(6); t_issue = time.perf_counter() - t0
loop.drain(); t_all = time.perf_counter() - t0
print("host: issue %.2f ms, complete %.2f ms for 6 segments (K = %d vector steps, 100 updates each)" % (t_issue * 1e3, t_all * 1e3, K))
for i, m in enumerate(loop.marks[n0:]):
    print("seg %d: " % i + "  ".join("%s %.2f" % (k, base.elapsed_time(m[k])) for k in ("r0", "r1", "l0", "l1", "c0", "c1")))
