"""Do the rollout's launches and the learner's launches overlap when issued on two streams?  Wall time of A alone, B alone, A || B."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import distributed_drl_amd as d
from distributed_drl_amd.agent import HyperParameters
from distributed_drl_amd.workers import RolloutDevice, TrainDevice
opt = HyperParameters(); opt.num_envs, opt.batch_size, opt.start_steps, opt.max_ep_len, opt.seed = 4096, 256, -1, 1000, 0
rb = d.ReplayBufferSAC1(8, 2, 1000000, seed=0)
n = 1000000
g = torch.Generator(device="cuda").manual_seed(1)
rb.store_batch(torch.randn(n, 8, device="cuda", generator=g), torch.rand(n, 2, device="cuda", generator=g), torch.randn(n, device="cuda", generator=g),
               torch.randn(n, 8, device="cuda", generator=g), torch.zeros(n, device="cuda"))
UPG = int(os.environ.get("UPG", "50"))
tr = TrainDevice(None, rb, opt, updates_per_graph=UPG)
ps = d.ParameterServer(*tr.agent.get_weights()); tr.ps = ps
st = d.ReplayBufferSAC1(8, 2, 100 * 4096)
ro = RolloutDevice(ps, st, opt)
ro.auto_pull = False
sR, sL = torch.cuda.Stream(), torch.cuda.Stream()
tr.run(2 * max(UPG, 1)); ro.step(5); torch.cuda.synchronize()
def wall(fa, fb, reps=5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        if fa:
            with torch.cuda.stream(sR): fa()
        if fb:
            with torch.cuda.stream(sL): fb()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
A = lambda: ro.step(100)
B = lambda: tr.run(100)
for _ in range(2):
    print("UPG %d: rollout 100 steps alone %.2f ms | learner 100 updates alone %.2f ms | both on two streams %.2f ms" % (UPG, wall(A, None), wall(None, B), wall(A, B)), flush=True)
# the same with the two halves of each kind beside each other
sR2 = torch.cuda.Stream()
def two_rollouts():
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        with torch.cuda.stream(sR): ro.step(100)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / 5 * 1e3
x = torch.randn(4096, 4096, device="cuda")
def mm():
    for _ in range(20): torch.mm(x, x)
print("20 torch.mm alone %.2f ms | rollout alone %.2f | mm || rollout %.2f" % (wall(None, mm), wall(A, None), wall(A, mm)))
print("20 torch.mm alone %.2f ms | learner alone %.2f | mm || learner %.2f" % (wall(mm, None), wall(None, B), wall(mm, B)))
