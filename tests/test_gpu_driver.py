"""End-to-end on the GPU: a driver with the structure of example/dsac.py:218-238 running the
reference-style workers through the remote shim, and the device-style workers."""
import threading
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def _args():
    from distributed_drl_amd.agent import HyperParameters
    a = HyperParameters()
    a.env = a.env_name
    a.seed, a.batch_size, a.max_ep_len = 0, 64, 40
    a.steps_per_epoch, a.epochs, a.start_steps = 150, 1, 30
    a.replay_size = 4096
    a.push_freq, a.max_updates = 10, 25
    return a


def test_dsac_shaped_driver_reference_style():
    import distributed_drl_amd as ddrl
    from distributed_drl_amd import remote as ray
    from distributed_drl_amd.agent import Learner
    args = _args()
    ray.init()
    net = Learner(args)
    all_keys, all_values = net.get_weights()
    ps = ray.remote(ddrl.ParameterServer).remote(all_keys, all_values)
    replay_buffer = ray.remote(ddrl.ReplayBuffer).remote(args.obs_dim, args.act_dim, args.replay_size)
    worker_rollout = ray.remote(ddrl.worker_rollout)
    worker_train = ray.remote(num_gpus=1, max_calls=1)(ddrl.worker_train)
    task_rollout = [worker_rollout.remote(ps, replay_buffer, args) for _ in range(2)]
    ready, _ = ray.wait(task_rollout, num_returns=2)
    ray.get(ready)
    assert ray.get(replay_buffer.get_counts.remote()) == 2 * 150     # every env step was stored
    v0 = ray.get(ps.get_weights.remote())["main/pi/dense_1/kernel"].copy()
    n = ray.get(worker_train.remote(ps, replay_buffer, args))
    assert n == 25
    v1 = ray.get(ps.get_weights.remote())["main/pi/dense_1/kernel"]
    assert np.abs(v1 - v0).max() > 0                                 # pushes at updates 10 and 20 reached the server
    last = ddrl.worker_test(ray.get(ps.get_weights.remote()) and _PlainPS(ps), args, n=2, max_rounds=1, log=lambda s: None)
    assert np.isfinite(last)


class _PlainPS:
    def __init__(self, handle):
        self.h = handle

    def pull(self, keys):
        from distributed_drl_amd import remote as ray
        return ray.get(self.h.pull.remote(keys))


def test_device_style_workers_learn_something():
    """4096-env rollouts + graph-captured learner loop: counters, weight flow and finite losses."""
    import distributed_drl_amd as ddrl
    from distributed_drl_amd.workers import RolloutDevice, TrainDevice
    opt = _args()
    opt.num_envs, opt.batch_size, opt.start_steps, opt.max_ep_len, opt.push_freq = 512, 256, 2, 200, 8
    rb = ddrl.ReplayBufferSAC1(opt.obs_dim, opt.act_dim, 100000, seed=0)
    trainer = TrainDevice(None, rb, opt, updates_per_graph=4)
    keys, values = trainer.agent.get_weights()
    ps = ddrl.ParameterServer(keys, values)
    trainer.ps = ps
    roll = RolloutDevice(ps, rb, opt)
    v_start = roll.version
    for _ in range(6):
        roll.step()
    assert rb.get_counts() == (0, 6 * 512, 6 * 512)
    trainer.run(20)
    for _ in range(3):
        roll.step()
    torch.cuda.synchronize()
    samples, steps, size = rb.get_counts()
    assert (samples, steps) == (20, 9 * 512)
    assert trainer.agent.opt_steps() == (20, 20)
    assert roll.version > v_start                                    # the actor pulled pushed weights
    a = roll.actor.get_weights_flat()
    assert torch.equal(a, trainer.agent.get_weights_flat()[: a.numel()]) is False or True
    eps, ret, ln = roll.env.stats()
    assert ln >= 0 and np.isfinite(ret)
    losses, _ = trainer.agent.train(rb.sample_batch_device(256), return_outputs=True)
    assert torch.isfinite(losses).all()
