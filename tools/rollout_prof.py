"""Rollout-only vector steps (policy forward + env.step + store, no learner) for rocprofv3 --kernel-trace --stats."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import distributed_drl_amd as d
from distributed_drl_amd.agent import HyperParameters
from distributed_drl_amd.workers import RolloutDevice

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
n_envs = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
opt = HyperParameters(num_workers=1)
opt.num_envs, opt.batch_size, opt.start_steps, opt.max_ep_len, opt.seed = n_envs, 256, -1, 1000, 0
rb = d.ReplayBufferSAC1(opt.obs_dim, opt.act_dim, 10 ** 6, seed=0)
roll = RolloutDevice(None, rb, opt, worker_index=0)
roll.step(20)
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
roll.step(n)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("rollout-only, %d envs: %.2f us per vector step, %.1f M env-steps/s" % (opt.num_envs, dt / n * 1e6, opt.num_envs * n / dt / 1e6))
