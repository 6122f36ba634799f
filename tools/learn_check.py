#!/usr/bin/env python3
"""Dev check: does the whole loop learn?  RolloutDevice (4096 envs) + TrainDevice (graph loop) + ParameterServer
through ActorLearnerLoop for `seconds`; prints the mean return of the episodes finished in each window."""
import sys
import time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import distributed_drl_amd as d
from distributed_drl_amd.agent import HyperParameters, Learner

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
opt = HyperParameters()
opt.num_envs, opt.start_steps, opt.max_ep_len, opt.seed = 4096, 5, 1000, 0
opt.lr = float(sys.argv[2]) if len(sys.argv) > 2 else 3e-4
if len(sys.argv) > 3:
    opt.alpha = float(sys.argv[3])
if len(sys.argv) > 4:
    opt.gamma = float(sys.argv[4])
if len(sys.argv) > 5:
    opt.num_envs = int(sys.argv[5])
if len(sys.argv) > 6:
    opt.start_steps = int(sys.argv[6])
print("lr %g alpha %g gamma %g envs %d start_steps %d" % (opt.lr, opt.alpha, opt.gamma, opt.num_envs, opt.start_steps), flush=True)
rb = d.ReplayBufferSAC1(opt.obs_dim, opt.act_dim, 10 ** 6, seed=0)
ps = d.ParameterServer(*Learner(opt).get_weights())
ro = d.RolloutDevice(ps, rb, opt)
tr = d.TrainDevice(ps, rb, opt, updates_per_graph=32)
loop = d.ActorLearnerLoop(ro, tr, opt)
t0 = time.time()
win = 0
while time.time() - t0 < seconds:
    loop.run(10)
    torch.cuda.synchronize()
    if time.time() - t0 > (win + 1) * seconds / 12:
        win += 1
        ep, ret, ln = ro.env.stats()
        print("t=%5.1fs  env-steps %9d  updates %8d  episodes %6d  mean return %9.2f  mean len %6.1f" %
              (time.time() - t0, loop.steps, loop.sample_times, ep, ret / max(ep, 1), ln / max(ep, 1)), flush=True)
