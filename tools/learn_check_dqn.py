#!/usr/bin/env python3
"""Dev check: do the Double-DQN / soft-Q learners learn?  A vector of lander envs driven through gym's discrete action set (0 nothing,
1 left engine, 2 main engine, 3 right engine, mapped onto the continuous stand-in's two controls), epsilon-greedy on the learner's own
q network (dqn: 0.97 greedy as actor_learner.py:193-201; sqn: a sample from softmax(q1 / alpha)), a ReplayBufferDQN, `updates_per_step`
updates per vector step — everything device-resident through the product's Python surface.
python tools/learn_check_dqn.py [seconds] [ddqn|sqn] [envs] [updates_per_step] [lr]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import distributed_drl_amd as d
from distributed_drl_amd import dqn
from distributed_drl_amd.env import VecLunarLander

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
variant = sys.argv[2] if len(sys.argv) > 2 else "ddqn"
n = int(sys.argv[3]) if len(sys.argv) > 3 else 256
ups = int(sys.argv[4]) if len(sys.argv) > 4 else 64


class Opt:
    obs_dim, act_dim, hidden_size, gamma, polyak, batch_size, seed, alpha = 8, 4, [400, 300], 0.99, 0.995, 256, 0, 0.2
    lr = float(sys.argv[5]) if len(sys.argv) > 5 else 3e-4
    buffer_size, save_dir = 1000000, "."


learner = (dqn.LearnerSQN if variant == "sqn" else dqn.Learner)(Opt, "learner")
rb = d.ReplayBufferDQN(Opt, 0, seed=0)
env = VecLunarLander(n, seed=0, max_ep_len=1000)
table = torch.tensor([[0.0, 0.0], [0.0, -1.0], [1.0, 0.0], [0.0, 1.0]], device="cuda")   # gym's discrete lander on the continuous controls
g = torch.Generator(device="cuda").manual_seed(0)
print("%s, %d envs, %d updates per vector step, lr %g, gamma %g" % (variant, n, ups, Opt.lr, Opt.gamma), flush=True)
t0, win, steps, updates = time.time(), 0, 0, 0
while time.time() - t0 < seconds:
    obs = env.obs.clone()
    q = learner.q_values(obs)
    if variant == "sqn":
        a = torch.multinomial(torch.softmax(q / Opt.alpha, 1), 1, generator=g)[:, 0]
    else:
        greedy = torch.rand(n, device="cuda", generator=g) < (0.97 if steps * n > 20000 else 0.0)
        a = torch.where(greedy, q.argmax(1), torch.randint(0, 4, (n,), device="cuda", generator=g))
    o2, r, dn, _, _ = env.step(table[a])
    rb.store_batch(obs, a.float(), r, o2, dn)
    steps += 1
    if steps * n > 5000:
        for _ in range(ups):
            learner.train(rb.sample_batch_device(Opt.batch_size), updates)
            updates += 1
    if time.time() - t0 > (win + 1) * seconds / 12:
        win += 1
        ep, ret, ln = env.stats()
        print("t=%5.1fs  env-steps %9d  updates %8d  episodes %6d  mean return %9.2f  mean len %6.1f" %
              (time.time() - t0, steps * n, updates, ep, ret / max(ep, 1), ln / max(ep, 1)), flush=True)
