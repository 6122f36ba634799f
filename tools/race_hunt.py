"""Race hunt: two identical learner runs (same ring, same seeds) must agree after EVERY update, also while a second process keeps
the GPU busy (timing perturbation: the configuration in which tests/test_partition.py first saw run-to-run differences).
    python tools/race_hunt.py [updates] [batch] [per_graph] [load: 0/1]
Prints the first update at which gradient / parameters / Adam state differ and which tensors the differing elements belong to."""
import os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
per_graph = int(sys.argv[3]) if len(sys.argv) > 3 else 0
load = int(sys.argv[4]) if len(sys.argv) > 4 else 1
if len(sys.argv) > 5 and sys.argv[5] == "child":      # the load generator: a learner loop of its own
    import numpy as np, torch
    import distributed_drl_amd as d
    from distributed_drl_amd.agent import HyperParameters
    from distributed_drl_amd.workers import TrainDevice
    opt = HyperParameters(); opt.batch_size, opt.seed, opt.push_freq = 256, 3, 10 ** 9
    rb = d.ReplayBufferSAC1(8, 2, 20000, seed=5)
    rb.store_batch(torch.randn(20000, 8).cuda(), torch.rand(20000, 2).cuda(), torch.randn(20000).cuda(), torch.randn(20000, 8).cuda(), torch.zeros(20000).cuda())
    td = TrainDevice(None, rb, opt, learner_index=0, updates_per_graph=0)
    t0 = time.time()
    while time.time() - t0 < float(sys.argv[6]):
        td.run(200); torch.cuda.synchronize()
    sys.exit(0)

import numpy as np
import torch
import distributed_drl_amd as d
from distributed_drl_amd import _lib
from distributed_drl_amd.agent import HyperParameters, param_specs
from distributed_drl_amd.workers import TrainDevice

child = None
if load:
    child = subprocess.Popen([sys.executable, os.path.abspath(__file__), "0", "0", "0", "0", "child", "60"])
    time.sleep(8)
names = []
opt0 = HyperParameters()
for k, sh in param_specs(opt0.obs_dim, opt0.act_dim, opt0.hidden_sizes[0], opt0.hidden_sizes[1], ("pi", "q1", "q2")):
    names += [(k, int(np.prod(sh)))]
def where(idx):
    off = 0
    for k, c in names:
        if idx < off + c:
            return "%s[%d]" % (k, idx - off)
        off += c
    return "?"
dumps = []
for rep in range(2):
    opt = HyperParameters(); opt.batch_size, opt.seed, opt.push_freq = B, 1, 10 ** 9
    rb = d.ReplayBufferSAC1(8, 2, 20000, seed=5)
    rs = np.random.RandomState(0); m = 20000
    rb.store_batch(*(torch.from_numpy(x).cuda() for x in (rs.randn(m, 8).astype(np.float32), rs.uniform(-1, 1, (m, 2)).astype(np.float32),
                                                           rs.randn(m).astype(np.float32), rs.randn(m, 8).astype(np.float32),
                                                           (rs.rand(m) < 0.01).astype(np.float32))))
    td = TrainDevice(None, rb, opt, learner_index=0, updates_per_graph=per_graph)
    step = max(1, per_graph)
    cur = []
    for u in range(0, n, step):
        td.run(step)
        cur.append([td.agent.export(w).cpu().numpy() for w in (_lib.SAC1_GRAD, _lib.SAC1_MAIN, _lib.SAC1_TARGET, _lib.SAC1_ADAM_M, _lib.SAC1_ADAM_V)])
    dumps.append(cur)
if child:
    child.kill(); child.wait()
bad = False
for u, (x, y) in enumerate(zip(*dumps)):
    for nm, a, b in zip(("grad", "main", "target", "m", "v"), x, y):
        ix = np.flatnonzero(a != b)
        if ix.size:
            bad = True
            print("update %d: %s differs in %d elements: first %s, last %s; tensors: %s" % (
                (u + 1) * max(1, per_graph), nm, ix.size, where(ix[0]), where(ix[-1]), sorted({where(i).split("[")[0] for i in ix[:: max(1, ix.size // 200)]})))
    if bad:
        break
print("race_hunt batch %d per_graph %d load %d: %s" % (B, per_graph, load, "DIFFERENT" if bad else "identical over %d updates" % n))
