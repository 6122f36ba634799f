#!/usr/bin/env python3
"""Checker-side helper (imports the oracles, hence under tests/): the long-horizon table of tests/_long_horizon.py (HIP learner beside the float32 and float64 oracles).
usage: python tests/_long_horizon_table.py [n_updates] [lr]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))  # _long_horizon lives beside this file
import numpy as np
import _long_horizon as lh
from distributed_drl_amd.agent import HyperParameters, Learner
from oracle import sac1_oracle as so

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
opt = HyperParameters()
opt.seed = 4
if len(sys.argv) > 2:
    opt.lr = float(sys.argv[2])
learner = Learner(opt)
cfg = so.Config(obs_dim=opt.obs_dim, act_dim=opt.act_dim, hidden1=opt.hidden_sizes[0], hidden2=opt.hidden_sizes[1], batch=opt.batch_size,
                alpha=opt.alpha, gamma=opt.gamma, lr=opt.lr, polyak=opt.polyak)
params = so.init_params(cfg, 4)
rs = np.random.RandomState(14)
for k in params:
    if k.endswith("bias"):
        params[k] = rs.uniform(-0.05, 0.05, params[k].shape).astype(np.float32)
print("n_updates %d lr %g alpha %g gamma %g polyak %g batch %d" % (n, cfg.lr, cfg.alpha, cfg.gamma, cfg.polyak, cfg.batch))
lh.report(lh.run(learner, cfg, params, n))
