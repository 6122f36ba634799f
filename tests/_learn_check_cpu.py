"""Helper (not collected): the SAC1 actor-learner loop on the CPU oracles ALONE — oracle/env_oracle.py's lander stand-in + oracle/sac1_oracle.py's
float32 learner, a NumPy replay ring, the reference's actor/learner gate (algos/sac1/sac1.py:203-207) and per-env weight adoption at episode ends
(example/dsac.py:127-130) — for the learning-curve control of DESIGN section 7: do the hyper-parameters bench.py carries (algos/sac1/hyperparams.py:60-82)
learn on an implementation that shares no code with the HIP loop, and what does gamma do there?  No HIP, no GPU.
usage: python tests/_learn_check_cpu.py --envs 32 --gamma 0.997 --hours 6"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from oracle import sac1_oracle as so
from oracle.env_oracle import LanderOracle


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=32)
    ap.add_argument("--lr", type=float, default=5e-5)
    ap.add_argument("--alpha", type=float, default=0.1)
    ap.add_argument("--gamma", type=float, default=0.997)
    ap.add_argument("--polyak", type=float, default=0.995)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--a-l-ratio", type=float, default=2.0)
    ap.add_argument("--push-freq", type=int, default=300)
    ap.add_argument("--start-env-steps", type=int, default=50000)
    ap.add_argument("--hours", type=float, default=6.0)
    ap.add_argument("--max-updates", type=int, default=10 ** 9)
    ap.add_argument("--window", type=int, default=20000, help="updates per report line")
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    torch.set_num_threads(1)
    N, B = a.envs, a.batch
    cfg = so.Config(batch=B, alpha=a.alpha, gamma=a.gamma, lr=a.lr, polyak=a.polyak)
    learner = so.Sac1Oracle(cfg, so.init_params(cfg, a.seed), torch.float32)
    env = LanderOracle(N, seed=a.seed, max_ep_len=1000)
    rs = np.random.RandomState(a.seed)
    cap = 10 ** 6
    ring = dict(obs1=np.zeros((cap, 8), np.float32), obs2=np.zeros((cap, 8), np.float32), acts=np.zeros((cap, 2), np.float32),
                rews=np.zeros(cap, np.float32), done=np.zeros(cap, np.float32))
    ptr = size = steps = samples = 0
    pi_of = lambda: {k: v.detach().numpy().copy() for k, v in learner.main.items() if "/pi/" in k}
    versions, env_ver = {0: pi_of()}, np.zeros(N, np.int64)
    newest = 0
    obs = env.obs()
    print("CPU oracles: envs %d batch %d a_l_ratio %g push_freq %d hidden (%d, %d) | lr %g alpha %g gamma %g polyak %g | random actions for the first %d env steps | seed %d"
          % (N, B, a.a_l_ratio, a.push_freq, cfg.hidden1, cfg.hidden2, a.lr, a.alpha, a.gamma, a.polyak, a.start_env_steps, a.seed), flush=True)
    t0 = time.time()
    next_report, best, u200 = a.window, -1e9, None
    while samples < a.max_updates and time.time() - t0 < a.hours * 3600:
        if steps > a.start_env_steps:
            act = np.empty((N, 2), np.float32)
            eps = rs.randn(N, 2).astype(np.float32)
            for v in np.unique(env_ver):
                m = env_ver == v
                act[m] = so.actor_act(cfg, versions[int(v)], obs[m], eps[m])
        else:
            act = rs.uniform(-1, 1, (N, 2)).astype(np.float32)
        o2, r, d, nxt, ended = env.step(act)
        idx = (ptr + np.arange(N)) % cap
        ring["obs1"][idx], ring["acts"][idx], ring["rews"][idx], ring["obs2"][idx], ring["done"][idx] = obs, act, r, o2, d
        ptr, size, steps = (ptr + N) % cap, min(cap, size + N), steps + N
        obs = nxt
        env_ver[ended.astype(bool)] = newest               # ps.pull at THIS env's episode end (dsac.py:127-130)
        due = int(steps // a.a_l_ratio) - samples if steps > a.start_env_steps else 0
        for _ in range(max(0, due)):
            ii = rs.randint(0, size, B)
            batch = {k: v[ii] for k, v in ring.items()}
            e = rs.randn(3, B, 2).astype(np.float32)
            learner.step(batch, e[0], e[1], e[2])
            samples += 1
            if samples % a.push_freq == 0:                 # ps.push (sac1.py:149)
                newest += 1
                versions[newest] = pi_of()
                for k in [k for k in versions if k != newest and not (env_ver == k).any()]:
                    del versions[k]
            if samples >= next_report:
                next_report += a.window
                ep, ret, ln = env.stats()
                mean = ret / max(ep, 1)
                best = max(best, mean) if ep else best
                if u200 is None and ep and mean >= 200.0:
                    u200 = samples
                print("t=%7.0fs  env-steps %9d  updates %8d  episodes %6d  mean return %9.2f  mean len %6.1f" %
                      (time.time() - t0, steps, samples, ep, mean, ln / max(ep, 1)), flush=True)
    print("best window %.2f; first window at >= 200: %s" % (best, "%d updates" % u200 if u200 is not None else "never"), flush=True)


if __name__ == "__main__":
    main()
