"""Two rollout workers (each 4096 / 8192 envs with its own replay shard, as the reference's several worker_rollout tasks and buffers) on
two streams of one GPU vs one worker with all the envs: the policy forward of one worker runs beside the env step of the other.
python tools/rollout_two.py [envs_per_worker]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import distributed_drl_amd as d
from distributed_drl_amd.agent import HyperParameters
from distributed_drl_amd.workers import RolloutDevice

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096


def mk(envs, idx):
    o = HyperParameters(num_workers=1)
    o.num_envs = envs
    o.start_steps = -1   # policy phase from the first step
    rb = d.ReplayBufferSAC1(o.obs_dim, o.act_dim, 1 << 20, seed=idx)
    return RolloutDevice(None, rb, o, worker_index=idx), rb


one, _ = mk(2 * n, 0)
a, _ra = mk(n, 1)
b, _rb = mk(n, 2)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()


def run_one(k):
    one.step(k)


def run_two(k):
    for _ in range(k):   # one vector step of each worker per turn, on its own stream
        with torch.cuda.stream(sa):
            a.step(1)
        with torch.cuda.stream(sb):
            b.step(1)


for name, fn in (("one worker, %d envs" % (2 * n), run_one), ("two workers x %d envs on two streams" % n, run_two)):
    fn(20); torch.cuda.synchronize(); t0 = time.perf_counter(); fn(400); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 400
    print("%-44s %.1f us per %d env steps = %.0f M env-steps/s" % (name, dt * 1e6, 2 * n, 2 * n / dt / 1e6))
