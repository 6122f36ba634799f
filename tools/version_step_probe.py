"""The vector step with several policy versions live (exact per-env adoption): k pushes between steps, then timed steps.
python tools/version_step_probe.py [pushes] [n_envs]   (under rocprofv3 --kernel-trace for the per-kernel split)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import distributed_drl_amd as d
from distributed_drl_amd.agent import HyperParameters
from distributed_drl_amd.workers import RolloutDevice, TrainDevice

pushes = int(sys.argv[1]) if len(sys.argv) > 1 else 16
n_envs = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
opt = HyperParameters(num_workers=1)
opt.num_envs, opt.batch_size, opt.start_steps, opt.max_ep_len, opt.seed = n_envs, 256, -1, 1000, 0
rb = d.ReplayBufferSAC1(opt.obs_dim, opt.act_dim, 10 ** 6, seed=0)
trainer = TrainDevice(None, rb, opt, learner_index=0, updates_per_graph=16)
keys, values = trainer.agent.get_weights()
ps = d.ParameterServer(keys, values)
roll = RolloutDevice(ps, rb, opt, worker_index=0)
roll.step(300)                       # episodes end at different steps from here on
for i in range(pushes):
    ps.push(keys, [v * (1.0 + 1e-3 * (i + 1)) for v in values])
    roll.step(); roll.step(3)
torch.cuda.synchronize()
_, vs = roll.actor.version_state(with_slots=False)
t0 = time.perf_counter()
for _ in range(10):
    roll.step(20)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 200
print("versions live %d, row tiles %d: %.2f us per vector step of %d envs" % (vs["live"], vs["tiles"], dt * 1e6, n_envs))
