"""Host cost of the parameter-server calls with NumPy values (example/dsac.py:59-65): get_weights, push, pull, set_weights.  python tools/ps_cost.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import distributed_drl_amd as d
from distributed_drl_amd.agent import HyperParameters, Learner, Actor
opt = HyperParameters(); opt.batch_size = 256
L = Learner(opt); A = Actor(opt)
keys, values = L.get_weights()
ps = d.ParameterServer(keys, values)
def timed(f, k=200, warm=20):
    for _ in range(warm): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / k * 1e6
print("types:", type(values[0]), len(keys), sum(int(np.prod(v.shape)) for v in values))
print("Learner.get_weights(): %.0f us" % timed(lambda: L.get_weights()))
print("ps.push(keys, values): %.0f us" % timed(lambda: ps.push(keys, values)))
print("ps.pull(keys): %.0f us" % timed(lambda: ps.pull(keys)))
w = ps.pull(keys)
print("pull types:", type(w[0]))
print("Actor.set_weights(keys, pulled): %.0f us" % timed(lambda: A.set_weights(keys, w)))
print("Learner.set_weights(keys, values): %.0f us" % timed(lambda: L.set_weights(keys, values)))
