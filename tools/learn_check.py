#!/usr/bin/env python3
"""Dev check: does the whole loop learn?  RolloutDevice (vectorised envs, per-env weight adoption) + TrainDevice (graph-captured learner
loop) + ParameterServer through ActorLearnerLoop (the reference's actor/learner gate, algos/sac1/sac1.py:203-207) for `--seconds`;
prints the mean return of the episodes that FINISHED in each window.  Defaults = bench.py's configuration (BASELINE config 2): 4096 envs,
batch 256, a_l_ratio 2, push every 300 updates, hidden (400, 300), and the SAC1 hyper-parameters of algos/sac1/hyperparams.py:60-82
(lr 5e-5, alpha 0.1, gamma 0.997, polyak 0.995, start_steps 5e4 env steps).  `--preset dsac` = example/dsac.py:193-204's values
(lr 1e-3, alpha 0.2, gamma 0.99, start_steps 1e4); `--preset lander` = the setting round 4 found to learn at 256 envs (lr 3e-4, alpha 0.2,
gamma 0.99)."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import distributed_drl_amd as d
from distributed_drl_amd.agent import HyperParameters, Learner

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=360.0)
ap.add_argument("--preset", choices=("sac1", "dsac", "lander"), default="sac1")
ap.add_argument("--envs", type=int, default=4096)
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--a-l-ratio", type=float, default=2.0)
ap.add_argument("--push-freq", type=int, default=300)
ap.add_argument("--lr", type=float, default=None)
ap.add_argument("--alpha", type=float, default=None)
ap.add_argument("--gamma", type=float, default=None)
ap.add_argument("--polyak", type=float, default=None)
ap.add_argument("--start-env-steps", type=int, default=None, help="random actions until this many env steps in all (the reference counts per worker)")
ap.add_argument("--max-ep-len", type=int, default=1000)
ap.add_argument("--windows", type=int, default=24)
ap.add_argument("--seed", type=int, default=0)
ap.add_argument("--free", type=int, default=0, metavar="K", help="free-running mode (workers.FreeRunningLoop: no gate, example/dsac.py:229-236): K vector steps "
                "per segment on the rollout stream beside --free-updates updates on the learner stream")
ap.add_argument("--free-updates", type=int, default=100)
args = ap.parse_args()

preset = {"sac1": dict(lr=5e-5, alpha=0.1, gamma=0.997, polyak=0.995, start=50000),
          "dsac": dict(lr=1e-3, alpha=0.2, gamma=0.99, polyak=0.995, start=10000),
          "lander": dict(lr=3e-4, alpha=0.2, gamma=0.99, polyak=0.995, start=10000)}[args.preset]
opt = HyperParameters(a_l_ratio=args.a_l_ratio)
opt.num_envs, opt.batch_size, opt.push_freq, opt.max_ep_len, opt.seed = args.envs, args.batch, args.push_freq, args.max_ep_len, args.seed
opt.lr = preset["lr"] if args.lr is None else args.lr
opt.alpha = preset["alpha"] if args.alpha is None else args.alpha
opt.gamma = preset["gamma"] if args.gamma is None else args.gamma
opt.polyak = preset["polyak"] if args.polyak is None else args.polyak
start_env = preset["start"] if args.start_env_steps is None else args.start_env_steps
opt.start_steps = max(0, -(-start_env // args.envs) - 1)        # RolloutDevice: random actions while t <= start_steps (vector steps)
print("preset %s: envs %d batch %d a_l_ratio %g push_freq %d hidden %s | lr %g alpha %g gamma %g polyak %g | random actions for %d vector steps "
      "(%d env steps), max_ep_len %d, seed %d" % (args.preset, opt.num_envs, opt.batch_size, opt.a_l_ratio, opt.push_freq, opt.hidden_sizes, opt.lr,
                                                  opt.alpha, opt.gamma, opt.polyak, opt.start_steps + 1, (opt.start_steps + 1) * opt.num_envs,
                                                  opt.max_ep_len, opt.seed), flush=True)
rb = d.ReplayBufferSAC1(opt.obs_dim, opt.act_dim, 10 ** 6, seed=opt.seed)
ps = d.ParameterServer(*Learner(opt).get_weights())
ro = d.RolloutDevice(ps, rb, opt)
tr = d.TrainDevice(ps, rb, opt, updates_per_graph=50)
t0 = time.time()
if args.free > 0:
    ro.step(opt.start_steps + 1)                                 # the random-action phase fills the ring first (nothing to train on before)

    class _Free(d.FreeRunningLoop):
        steps = property(lambda self: self.env_steps + (opt.start_steps + 1) * opt.num_envs)
        sample_times = property(lambda self: self.updates)
    loop = _Free(ro, tr, opt, steps_per_segment=args.free, updates_per_segment=args.free_updates)
    run_some = lambda: (loop.run(8), loop.drain())
else:
    loop = d.ActorLearnerLoop(ro, tr, opt)
    loop.start_steps = (opt.start_steps + 1) * opt.num_envs          # the learner starts when the policy does (sac1.py:196)
    run_some = lambda: (loop.run(4 if args.envs >= 1024 else 32), torch.cuda.synchronize())
win, best, t200 = 0, -1e9, None
while time.time() - t0 < args.seconds:
    run_some()
    if time.time() - t0 > (win + 1) * args.seconds / args.windows:
        win += 1
        ep, ret, ln = ro.env.stats()
        mean = ret / max(ep, 1)
        best = max(best, mean) if ep else best
        if t200 is None and ep and mean >= 200.0:
            t200 = time.time() - t0
        print("t=%6.1fs  env-steps %10d  updates %9d  vector steps %6d  episodes %6d  mean return %9.2f  mean len %6.1f" %
              (time.time() - t0, loop.steps, loop.sample_times, loop.steps // opt.num_envs, ep, mean, ln / max(ep, 1)), flush=True)
print("best window %.2f; first window at >= 200: %s" % (best, "%.0f s" % t200 if t200 is not None else "never"), flush=True)
