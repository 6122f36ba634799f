import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import distributed_drl_amd as d
from distributed_drl_amd.agent import HyperParameters, Learner
opt = HyperParameters(); B = 256; opt.batch_size = B
rbh = d.ReplayBufferSAC1(8, 2, 100000, seed=5)
rs = np.random.RandomState(0)
oo, aa = rs.randn(8), rs.uniform(-1, 1, 2).astype(np.float32)
for _ in range(200): rbh.store(oo, aa, 0.5, oo, False)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(2000): rbh.store(oo, aa, 0.5, oo, False)
torch.cuda.synchronize(); t_store = (time.perf_counter() - t0) / 2000
m = 100000
rbh.store_batch(torch.randn(m, 8).cuda(), torch.rand(m, 2).cuda(), torch.randn(m).cuda(), torch.randn(m, 8).cuda(), torch.zeros(m).cuda())
for _ in range(20): bh = rbh.sample_batch(B)
t0 = time.perf_counter()
for _ in range(300): bh = rbh.sample_batch(B)
t_samp = (time.perf_counter() - t0) / 300
lh = Learner(opt, job="learner", index=77)
for _ in range(5): lh.train(bh)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(300): lh.train(rbh.sample_batch(B))
torch.cuda.synchronize(); t_iter = (time.perf_counter() - t0) / 300
print("store %.1f us  sample_batch %.1f us  sample+train %.1f us" % (t_store * 1e6, t_samp * 1e6, t_iter * 1e6))
# the reference's Cache inside the buffer (ReplayBuffer.prefetch): ten draws always in flight, sample_batch hands out the oldest
rbh.prefetch(B)
for _ in range(20): lh.train(rbh.sample_batch(B))
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(300): bh = rbh.sample_batch(B)
t_samp_p = (time.perf_counter() - t0) / 300
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(1000): lh.train(rbh.sample_batch(B))
torch.cuda.synchronize(); t_iter_p = (time.perf_counter() - t0) / 1000
rbh.prefetch(0)
print("with ReplayBuffer.prefetch: sample_batch %.1f us  sample+train %.1f us = %.0f updates/s" % (t_samp_p * 1e6, t_iter_p * 1e6, 1.0 / t_iter_p))
rbh.prefetch(B, own_stream=True)
for _ in range(20): lh.train(rbh.sample_batch(B))
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(1000): lh.train(rbh.sample_batch(B))
torch.cuda.synchronize(); t_iter_o = (time.perf_counter() - t0) / 1000
rbh.prefetch(0)
print("with ReplayBuffer.prefetch(own_stream=True): sample+train %.1f us = %.0f updates/s" % (t_iter_o * 1e6, 1.0 / t_iter_o))
# the reference's own remedy for that loop: worker_train behind the Cache helper (algos/sac1/sac1.py:103-154) — replay buffer and
# parameter server as actors (remote.py: a thread + a HIP stream each), the helper drawing batch i + 1 while update i trains
from distributed_drl_amd import remote as ray
opt.max_updates, opt.push_freq = 3000, 300
for cached in (False, True, "prefetch", False, True, "prefetch"):
    ps_a = ray.remote(d.ParameterServer).remote(*lh.get_weights())
    rb_a = ray.remote(d.ReplayBufferSAC1).remote(8, 2, 100000, None, 5)
    ray.get(rb_a.store_batch.remote(torch.randn(m, 8).cuda(), torch.rand(m, 2).cuda(), torch.randn(m).cuda(), torch.randn(m, 8).cuda(), torch.zeros(m).cuda()))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n_upd = d.worker_train_sac1(ps_a, rb_a, opt, 0, make_agent=lambda o_: lh, make_cache={True: None, False: False, "prefetch": "prefetch"}[cached])
    torch.cuda.synchronize(); t_w = (time.perf_counter() - t0) / n_upd
    print("worker_train_sac1 over actor handles, %-22s: %.1f us per update = %.0f updates/s" %
          ({True: "Cache helper thread", False: "no helper", "prefetch": "prefetch in the buffer"}[cached], t_w * 1e6, 1.0 / t_w))
# per-step policy calls of the reference-style rollout workers (one observation up, one action down)
from distributed_drl_amd.agent import Actor
from distributed_drl_amd import dqn
act = Actor(opt, job="worker")
o = rs.randn(8)
for _ in range(50): act.get_action(o)
t0 = time.perf_counter()
for _ in range(1000): act.get_action(o)
t_sac = (time.perf_counter() - t0) / 1000


class O5:
    obs_dim, act_dim, hidden_size, gamma, lr, polyak, batch_size, seed, alpha = 38, 3, [400, 300], 0.99, 1e-3, 0.995, 128, 0, 0.1


da = dqn.Actor(O5, "worker")
o38 = rs.randn(38)
for _ in range(50): da.get_action(o38)
t0 = time.perf_counter()
for _ in range(1000): da.get_action(o38)
t_dqn = (time.perf_counter() - t0) / 1000
print("Actor.get_action %.1f us   dqn.Actor.get_action %.1f us" % (t_sac * 1e6, t_dqn * 1e6))
