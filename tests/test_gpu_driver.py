"""End-to-end on the GPU: a driver with the structure of example/dsac.py:218-238 running the
reference-style workers through the remote shim, and the device-style workers."""
import os
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
    assert torch.equal(a, ps.pull_flat(0, a.numel()))                              # the actor runs what the server holds: the push after update 16
    assert not torch.equal(a, trainer.agent.get_weights_flat()[: a.numel()])      # ... which the learner has moved on from (updates 17-20)
    eps, ret, ln = roll.env.stats()
    assert ln >= 0 and np.isfinite(ret)
    losses, _ = trainer.agent.train(rb.sample_batch_device(256), return_outputs=True)
    assert torch.isfinite(losses).all()


def test_nstep_rollout_equals_per_env_reference_loop():
    """RolloutDeviceNStep (Wrapper'd vector env + device window queues + masked window store) stores, for
    every env, exactly the windows the reference's worker loop (algos/sac1/sac_ray.py:192-262: deques of
    maxlen Ln+1 / Ln, store once t_queue >= Ln, new deques at episode end) would store for that env, and
    in env order within a vector step — ring contents and counters bit-exact vs a Python restatement
    with real deques over the NumPy env oracle."""
    from collections import deque
    import distributed_drl_amd as d
    from distributed_drl_amd.agent import HyperParameters
    from oracle.env_oracle import LanderOracle
    from oracle.replay_oracle import NStepReplayOracle
    from oracle import noise_oracle as no
    opt = HyperParameters()
    opt.num_envs, opt.seed, opt.Ln, opt.max_ep_len, opt.action_repeat = 24, 3, 4, 60, 2
    opt.buffer_size, opt.batch_size, opt.num_buffers = 150, 8, 1
    opt.obs_noise, opt.act_noise, opt.reward_scale = 0.01, 0.3, 5
    rb = d.ReplayBufferNStep(opt)
    ro = d.RolloutDeviceNStep(None, rb, opt)
    # ---- reference-order restatement: one deque pair per env, sequential over envs inside a step
    n, Ln = opt.num_envs, opt.Ln
    env = LanderOracle(n, seed=opt.seed, max_ep_len=1 << 23)
    ora = NStepReplayOracle(opt)
    oq = [deque([(env.obs()[e].copy(),)], maxlen=Ln + 1) for e in range(n)]
    aq = [deque([], maxlen=Ln) for e in range(n)]
    tq = [1] * n
    limit = -(-opt.max_ep_len // opt.action_repeat)
    steps = 70
    for t in range(steps):
        ro.step()
        act = no.uniform_fill(n * 2, -1.0, 1.0, opt.seed ^ 0x5EED5EED, counter=t * n * 2).reshape(n, 2)  # env.action_space.sample()
        o2, r, dd, nxt, ended = env.step_wrapped(act, opt.act_noise, opt.obs_noise, opt.reward_scale, 3, limit)
        np.testing.assert_array_equal(ro.act.cpu().numpy(), act)   # both now hold the noisy action (`action +=` in Wrapper.step)
        for e in range(n):
            aq[e].append((act[e].copy(), r[e], dd[e]))
            oq[e].append((o2[e].copy(),))
            if tq[e] >= Ln and tq[e] % opt.save_freq == 0:
                ora.store(oq[e], aq[e], 0)
            tq[e] += 1
            if ended[e]:
                oq[e] = deque([(nxt[e].copy(),)], maxlen=Ln + 1)
                aq[e] = deque([], maxlen=Ln)
                tq[e] = 1
    rings = rb.rings()
    for k in ("buffer_o", "buffer_a", "buffer_r", "buffer_d"):
        np.testing.assert_array_equal(rings[k].cpu().numpy(), getattr(ora, k), err_msg=k)
    assert rb.get_counts() == ora.get_counts()
    assert ora.steps > opt.buffer_size  # the ring wrapped


def test_actor_learner_ratio_gate():
    """sac1.py:203-207: once learning has started the rollouts never run ahead of
    steps / sample_times <= a_l_ratio; nothing is trained while the buffer fills (steps <= start_steps)."""
    import distributed_drl_amd as d
    from distributed_drl_amd.agent import HyperParameters
    opt = HyperParameters()
    opt.num_envs, opt.start_steps, opt.a_l_ratio, opt.batch_size, opt.hidden_sizes, opt.push_freq = 64, 256, 2, 32, (64, 32), 50
    rb = d.ReplayBufferSAC1(opt.obs_dim, opt.act_dim, 10000, seed=0)
    from distributed_drl_amd.agent import Learner
    net = Learner(opt)
    ps = d.ParameterServer(*net.get_weights())
    loop = d.ActorLearnerLoop(d.RolloutDevice(ps, rb, opt), d.TrainDevice(ps, rb, opt, updates_per_graph=4), opt)
    loop.run(4)                                   # 256 steps: still filling
    assert loop.counts() == (0, 256, 256)
    v0 = ps.version
    loop.run(20)
    samples, steps, size = loop.counts()
    assert steps == 24 * 64 and samples == steps // 2 and steps / samples <= opt.a_l_ratio
    assert (loop.steps, loop.sample_times) == (steps, samples)
    assert ps.version > v0                        # the learner pushed (every 50th update)


def test_free_running_loop_two_streams_equal_the_one_stream_order():
    """example/dsac.py:229-236 leaves rollouts and learner running with no gate between them.  FreeRunningLoop runs the vectorised
    rollout and the learner's graph loop on TWO streams, exchanging transitions (staging ring -> commit) and weights (pull of the
    learner's previous segment) at segment boundaries only.  Nothing inside a segment depends on the other stream, so the run must
    equal — bit for bit: parameters, ring contents, counters, sampler state, the envs — the same launches issued on ONE stream in
    the order [pull, K vector steps, n updates, commit]; and it is reproducible run to run."""
    import distributed_drl_amd as d
    from distributed_drl_amd.agent import HyperParameters, Learner

    def run_it(one_stream):
        opt = HyperParameters()
        opt.num_envs, opt.start_steps, opt.batch_size, opt.hidden_sizes, opt.push_freq, opt.max_ep_len, opt.seed = 64, -1, 32, (64, 32), 6, 40, 3
        rb = d.ReplayBufferSAC1(opt.obs_dim, opt.act_dim, 2000, seed=5)
        rs = np.random.RandomState(1)
        m = 300
        rb.store_batch(*(torch.from_numpy(x).cuda() for x in (
            rs.randn(m, 8).astype(np.float32), rs.uniform(-1, 1, (m, 2)).astype(np.float32), rs.randn(m).astype(np.float32),
            rs.randn(m, 8).astype(np.float32), np.zeros(m, np.float32))))
        net = Learner(opt)
        ps = d.ParameterServer(*net.get_weights())
        roll, tr = d.RolloutDevice(ps, rb, opt), d.TrainDevice(ps, rb, opt, updates_per_graph=4)
        st = torch.cuda.current_stream()
        loop = d.FreeRunningLoop(roll, tr, opt, steps_per_segment=5, updates_per_segment=9, streams=(st, st) if one_stream else None)
        loop.run(7)                                  # 35 vector steps (2240 transitions: the 2000-row ring wraps), 63 updates, pushes at 6, 12, ...
        loop.close()
        assert (loop.env_steps, loop.updates) == (7 * 5 * 64, 63)
        assert rb.get_counts() == (63, m + 7 * 5 * 64, 2000)
        assert ps.version >= 10 and roll.rb is rb
        rings = {k: v.clone() for k, v in rb.rings().items()}
        key, pos = rb.mt_state()
        return tr.agent.get_weights_flat().cpu().numpy(), rings, (key.copy(), pos), roll.env.obs.clone(), roll.actor.get_weights_flat().cpu().numpy()

    a, b, c = run_it(False), run_it(True), run_it(False)
    for x, y in ((a, b), (a, c)):
        np.testing.assert_array_equal(x[0], y[0])
        for k in x[1]:
            assert torch.equal(x[1][k], y[1][k]), k
        np.testing.assert_array_equal(x[2][0], y[2][0])
        assert x[2][1] == y[2][1] and torch.equal(x[3], y[3])
        np.testing.assert_array_equal(x[4], y[4])


def test_worker_train_sac1_behind_the_cache_equals_the_plain_loop():
    """algos/sac1/sac1.py:103-154: the learner takes its batches from the Cache helper's queue (q1, depth 10) and leaves its weights in
    q2.  One buffer, one FIFO queue: the learner trains on exactly the batches, in exactly the order, of `batch = sample_batch();
    train(batch)` without the helper — the server ends on bit-identical weights (pushes after updates 16 and 32)."""
    import distributed_drl_amd as d
    from distributed_drl_amd import remote as ray
    from distributed_drl_amd.agent import HyperParameters, Learner

    def run_it(cached):
        opt = HyperParameters()
        opt.batch_size, opt.hidden_sizes, opt.push_freq, opt.max_updates, opt.seed = 64, (64, 32), 16, 40, 2
        ray.init()
        net = Learner(opt)
        ps = ray.remote(d.ParameterServer).remote(*net.get_weights())
        rb = ray.remote(d.ReplayBufferSAC1).remote(opt.obs_dim, opt.act_dim, 5000, None, 11)
        rs = np.random.RandomState(4)
        for _ in range(200):
            rb.store.remote(rs.randn(8), rs.uniform(-1, 1, 2), float(rs.randn()), rs.randn(8), bool(rs.rand() < 0.1))
        n = d.worker_train_sac1(ps, rb, opt, 0) if cached else d.worker_train_sac1(ps, rb, opt, 0, make_cache=False)
        assert n == 40
        w = ray.get(ps.get_weights.remote())
        samples, steps, size = ray.get(rb.get_counts.remote())
        assert steps == 200 and (samples == 40 if not cached else 40 <= samples <= 40 + 12)   # the helper had drawn ahead
        return w

    a, b = run_it(True), run_it(False)
    assert list(a.keys()) == list(b.keys())
    for k in a:
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)


def test_checkpoint_resume_formats(tmp_path):
    """N3: (a) weights pickle {name: ndarray} + per-node recover (algos/dqn/train.py:111-174);
    (b) full learner state: a learner restored from save_state() continues bit-identically."""
    import pickle
    import distributed_drl_amd as d
    from distributed_drl_amd import _lib
    from distributed_drl_amd.agent import HyperParameters, Learner
    from oracle import sac1_oracle as so
    opt = HyperParameters()
    opt.hidden_sizes, opt.batch_size, opt.save_dir, opt.recover, opt.seed = (64, 32), 32, str(tmp_path / "run"), False, 4
    net = Learner(opt)
    keys, vals = net.get_weights()
    ps = d.ParameterServerNode(opt, "", "", 0, keys, vals)
    assert (tmp_path / "run" / "All_Parameters.json").exists()
    vals2 = [v + 1.0 for v in vals]
    ps.push(keys, vals2)
    assert ps.learner_step == opt.push_freq
    ps.save_weights()
    w = pickle.load(open(tmp_path / "run" / "checkpoint" / "checkpoint_weights.pickle", "rb"))
    assert list(w.keys()) == keys and all((w[k] == v).all() for k, v in zip(keys, vals2))
    opt.recover = True
    ps2 = d.ParameterServerNode(opt, "", "", 0, keys, vals)           # recovers the pushed weights, not `vals`
    for a, b in zip(ps2.pull(keys), vals2):
        np.testing.assert_array_equal(a, b)
    # ---- (b)
    cfg = so.Config(obs_dim=opt.obs_dim, act_dim=opt.act_dim, hidden1=64, hidden2=32, batch=32)
    l1 = Learner(opt)
    for it in range(3):
        batch, eps = so.synthetic_batch(cfg, seed=it)
        l1.train(batch, eps=eps)
    l1.save_state(str(tmp_path / "learner"))
    l2 = Learner(opt)
    l2.load_state(str(tmp_path / "learner"))
    assert l2.opt_steps() == (3, 3)
    for it in range(3, 6):
        batch, eps = so.synthetic_batch(cfg, seed=it)
        a, _ = l1.train(batch, eps=eps, return_outputs=True)
        b, _ = l2.train(batch, eps=eps, return_outputs=True)
        assert torch.equal(a, b)
    for which in (_lib.SAC1_MAIN, _lib.SAC1_TARGET, _lib.SAC1_ADAM_M, _lib.SAC1_ADAM_V):
        assert torch.equal(l1.export(which), l2.export(which))


def test_dsac_driver_with_its_own_model():
    """example/dsac.py:218-238 end to end with its own `args` (ac_kwargs, batch 100, lr 1e-3, alpha 0.2) and its
    own algorithm (example/model.py's SAC-v): ParameterServer / ReplayBuffer actors, one rollout task, one
    learner task, weights flow learner -> ps -> rollout."""
    import distributed_drl_amd as d
    from distributed_drl_amd import remote as ray
    from distributed_drl_amd.agent import Model

    class Args:
        env, obs_dim, act_dim = "LunarLanderContinuous-v2", 8, 2
        ac_kwargs = dict(hidden_sizes=[64, 64])
        gamma = 0.99,
        polyak, lr, alpha, batch_size, seed = 0.995, 1e-3, 0.2, 100, 0
        replay_size, start_steps, max_ep_len, steps_per_epoch, epochs = 5000, 150, 200, 400, 1
        push_freq, max_updates = 10, 40
    args = Args()
    ray.init()
    net = Model(args)
    all_keys, all_values = net.get_weights()
    assert len(all_keys) == 26
    ps = ray.remote(d.ParameterServer).remote(all_keys, all_values)
    rb = ray.remote(d.ReplayBuffer).remote(args.obs_dim, args.act_dim, args.replay_size)
    t_roll = ray.remote(d.worker_rollout).remote(ps, rb, args)
    ray.get(t_roll)                                              # 400 env steps stored
    assert ray.get(rb.get_counts.remote()) == 400
    v0 = ray.get(ps.pull.remote(all_keys))
    n = ray.get(ray.remote(d.worker_train).remote(ps, rb, args))  # 40 SAC-v updates, 4 pushes
    assert n == 40
    v1 = ray.get(ps.pull.remote(all_keys))
    assert any((a != b).any() for a, b in zip(v0, v1))
    assert all(np.isfinite(b).all() for b in v1)


def test_worker_test_sac1_saves_best_weights(tmp_path):
    """algos/sac1/sac1.py:214-252: deterministic test episodes, counters, save_weights on a new best return."""
    import pickle
    import distributed_drl_amd as d
    from distributed_drl_amd.agent import HyperParameters, Learner
    opt = HyperParameters()
    opt.hidden_sizes, opt.max_ep_len, opt.save_dir, opt.env_name = (64, 32), 60, str(tmp_path), "LunarLanderContinuous-v2"
    opt.summary_dir = str(tmp_path / "tb")          # the reference's tf.summary scalar (actor_learner.py:210-229)
    keys, vals = Learner(opt).get_weights()
    ps = d.ParameterServer(keys, vals)
    rb = d.ReplayBufferSAC1(opt.obs_dim, opt.act_dim, 100)
    lines = []
    best = d.worker_test_sac1(ps, rb, opt, log=lines.append, sleep=lambda s: None, max_rounds=2, n=2)
    assert best > -1000 and any("weights saved" in l for l in lines) and any(l.startswith("test_reward:") for l in lines)
    import glob
    from distributed_drl_amd import logx
    runs = glob.glob(str(tmp_path / "tb" / "*-LunarLanderContinuous-v2-workers_num:1%2"))   # one run directory, named as actor_learner.py:177-179 names it
    assert len(runs) == 1
    ev = logx.read_scalars(glob.glob(runs[0] + "/events.out.tfevents.*")[0])
    assert len(ev) == 2 and all(tag == "Reward" and step == 0 for step, tag, _ in ev)   # one scalar per round, at step sample_times
    w = pickle.load(open(tmp_path / "weights.pickle", "rb"))
    assert set(w.keys()) == set(keys)


def test_fused_rollout_step_matches_the_unfused_sequence_and_the_oracles():
    """ddrl_rollout_step (policy forward launch + one launch for get_action / env.step / store) vs the three-call sequence
    it replaces, from identical state and the same noise stream: actions within float32 rounding of each other and of the
    float64 oracle; given the GPU's own actions the env transition and the stored ring rows are BIT-exact vs the NumPy env
    oracle (the physics and the store are integer / exact-float work); counters, cursor and episode statistics agree."""
    import distributed_drl_amd as ddrl
    from distributed_drl_amd.agent import HyperParameters
    from distributed_drl_amd.workers import RolloutDevice
    from oracle import sac1_oracle as so
    from oracle.env_oracle import LanderOracle
    n = 4096   # BASELINE config 2
    opt = HyperParameters()
    opt.num_envs, opt.start_steps, opt.max_ep_len, opt.seed = n, -1, 40, 11
    from distributed_drl_amd.agent import Learner
    keys, vals = Learner(opt).get_weights()
    ps = ddrl.ParameterServer(keys, vals)
    rbs = [ddrl.ReplayBufferSAC1(8, 2, 3 * n + 100, seed=0) for _ in range(2)]
    opt_u = HyperParameters()
    opt_u.__dict__.update(opt.__dict__)
    opt_u.fused_rollout = False
    fused, plain = RolloutDevice(ps, rbs[0], opt), RolloutDevice(ps, rbs[1], opt_u)
    assert fused._fused_ready() and not plain._fused_ready()
    ora = LanderOracle(n, seed=opt.seed, max_ep_len=opt.max_ep_len)
    cfg = so.Config()
    params = dict(zip(keys, vals))
    for t in range(5):   # 5 x 4096 stores into a 12388-row ring: wraps; max_ep_len 40 is not reached, crashes do end episodes
        obs = fused.env.obs.cpu().numpy().copy()
        np.testing.assert_array_equal(obs, ora.obs())
        ctr = fused.actor._noise_ctr
        fused.step()
        plain.step()
        act = fused.act.cpu().numpy()
        # same policy, same eps stream: the two GPU paths differ only in the order of float32 sums
        np.testing.assert_allclose(act, plain.act.cpu().numpy(), rtol=0, atol=2e-6)
        eps = torch.empty(n * 2, device="cuda")
        from distributed_drl_amd import _lib
        _lib.check(_lib.load().ddrl_normal_fill(_lib.dptr(eps), n * 2, fused.actor._noise_seed, ctr, _lib.stream_ptr()))
        want = so.actor_act(cfg, params, obs, eps.view(n, 2).cpu().numpy(), dtype=torch.float64)
        np.testing.assert_allclose(act, want, rtol=1e-5, atol=2e-6)
        # the env + store half, bit for bit, given the GPU's own actions
        o2, r, d, nxt, ended = ora.step(act)
        np.testing.assert_array_equal(fused.env.obs.cpu().numpy(), nxt)
        rings = rbs[0].rings()
        cap = 3 * n + 100
        rows = (t * n + np.arange(n)) % cap
        keep = np.ones(n, bool)
        for k, w in (("obs1_buf", obs), ("obs2_buf", o2), ("acts_buf", act), ("rews_buf", r), ("done_buf", d)):
            np.testing.assert_array_equal(rings[k].cpu().numpy()[rows][keep], np.asarray(w, np.float32)[keep], err_msg="%s step %d" % (k, t))
        plain.env.set_state(fused.env.get_state())   # keep the two env sets identical (their actions differ in the last bits)
        plain.env.obs.copy_(fused.env.obs)
    assert rbs[0].get_counts() == rbs[1].get_counts() == (0, 5 * n, 3 * n + 100)
    assert rbs[0].ptr == (5 * n) % (3 * n + 100)
    assert fused.env.stats()[0] == ora.episodes


def test_actor_learner_loop_at_config2_sizes_counters_and_mt_state():
    """BASELINE config 2 exactly: 4096 envs, 10^6-transition ring, batch 256, a_l_ratio 2 -> every vector step owes 2048
    updates (graph-captured, 50 per graph, the sampler of update u+1 riding inside update u).  After three vector steps the
    device counters, the ring cursor and the MT19937 state of the index stream equal the oracle's after the same sequence of
    store(4096) / 2048 x sample_batch(256) calls — bit for bit (the index stream depends on the ring size at every draw)."""
    import distributed_drl_amd as ddrl
    from distributed_drl_amd.agent import HyperParameters, Learner
    from distributed_drl_amd.workers import ActorLearnerLoop, RolloutDevice, TrainDevice
    from oracle.replay_oracle import ReplayBufferOracle
    opt = HyperParameters()
    opt.num_envs, opt.batch_size, opt.start_steps, opt.a_l_ratio, opt.push_freq, opt.seed = 4096, 256, 0, 2, 300, 2
    cap, pre = 10 ** 6, 300000
    rb = ddrl.ReplayBufferSAC1(opt.obs_dim, opt.act_dim, cap, seed=4)
    ora = ReplayBufferOracle(opt.obs_dim, opt.act_dim, cap, seed=4)
    rs = np.random.RandomState(0)
    o, a, r = rs.randn(pre, 8).astype(np.float32), rs.uniform(-1, 1, (pre, 2)).astype(np.float32), rs.randn(pre).astype(np.float32)
    d = np.zeros(pre, np.float32)
    rb.store_batch(*(torch.from_numpy(x).cuda() for x in (o, a, r, o, d)))
    ora.ptr = ora.size = ora.steps = pre              # (the oracle's per-row Python store loop is not what is under test here)
    keys, vals = Learner(opt).get_weights()
    ps = ddrl.ParameterServer(keys, vals)
    loop = ActorLearnerLoop(RolloutDevice(ps, rb, opt), TrainDevice(ps, rb, opt, updates_per_graph=50), opt)
    for step in range(3):
        loop.run(1)
        ora.ptr, ora.size, ora.steps = ora.ptr + 4096, ora.size + 4096, ora.steps + 4096   # store(4096): contents do not matter for the index stream, sizes do
        for _ in range(2048):
            ora.sample_batch(256)
    torch.cuda.synchronize()
    assert loop.counts() == ora.get_counts() == (3 * 2048, pre + 3 * 4096, pre + 3 * 4096)
    assert rb.ptr == ora.ptr
    key, pos = rb.mt_state()
    assert pos == ora.rng.pos and (np.asarray(key) == ora.rng.key).all()
    assert loop.trainer.agent.opt_steps() == (3 * 2048, 3 * 2048)
    assert ps.version >= 1 + (3 * 2048) // 300


def test_dp_stepper_ride_along_sampler_equals_plain_sequence():
    """Data-parallel learner iteration (partition.py, config 4): drawing the next batch inside the current update's forward
    launch (ddrl_sac1_compute_grads_and_sample) trains on exactly the batches, in exactly the order, of draw -> gradients ->
    apply one at a time — parameters, targets and Adam moments bit for bit, and the sampler ends in the same state."""
    import distributed_drl_amd as ddrl
    from distributed_drl_amd import _lib
    from distributed_drl_amd.agent import HyperParameters, Learner
    opt = HyperParameters()
    opt.batch_size, opt.seed = 256, 4
    rs = np.random.RandomState(3)
    n = 6000
    data = [rs.randn(n, 8).astype(np.float32), rs.uniform(-1, 1, (n, 2)).astype(np.float32), rs.randn(n).astype(np.float32),
            rs.randn(n, 8).astype(np.float32), (rs.rand(n) < 0.05).astype(np.float32)]
    outs = []
    for ride in (False, True):
        rb = ddrl.ReplayBufferSAC1(8, 2, 8192, seed=21)
        rb.store_batch(*(torch.from_numpy(x).cuda() for x in data))
        L = Learner(opt, job="learner", index=0)
        grads, apply, g = L.dp_stepper(rb)
        for step in range(3):            # three "steps" of five updates: nothing is drawn ahead across a step's end
            for u in range(5):
                grads(last=(u == 4) or not ride)
                g.mul_(1.0)              # (where the in-place all-reduce goes)
                apply()
        outs.append(([L.export(w).clone() for w in (_lib.SAC1_MAIN, _lib.SAC1_TARGET, _lib.SAC1_ADAM_M, _lib.SAC1_ADAM_V)],
                     rb.get_counts(), rb.mt_state(), L.opt_steps()))
    for a, b in zip(outs[0][0], outs[1][0]):
        assert torch.equal(a, b)
    assert outs[0][1] == outs[1][1] == (15, n, n) and outs[0][3] == outs[1][3] == (15, 15)
    assert outs[0][2][1] == outs[1][2][1] and (outs[0][2][0] == outs[1][2][0]).all()


def test_vectorised_rollout_never_acts_on_weights_older_than_the_reference_worker_would():
    """DESIGN §7's claim, checked: the reference worker pulls at ITS episode end (example/dsac.py:129-130); the vectorised
    rollout adopts a push for ALL envs at the end of the first vector step after it.  Pushes land between vector steps;
    every policy version makes a recognisable action (zero kernels, bias-coded mean, log_std at its floor), so the ring rows
    tell which version each env acted on.  For every env and step: version used >= the version that env's last episode end
    would have pulled — and it is exactly the newest version pushed at least two vector steps ago."""
    import distributed_drl_amd as ddrl
    from distributed_drl_amd.agent import HyperParameters, Learner
    from distributed_drl_amd.workers import RolloutDevice
    n, steps = 64, 40
    opt = HyperParameters()
    opt.num_envs, opt.start_steps, opt.max_ep_len, opt.seed = n, -1, 7, 3   # time limit 7: every env ends episodes all the time
    opt.adopt = "step"                                                    # the whole-vector swap; "episode" (default): the tests below
    keys, vals = Learner(opt).get_weights()
    code = lambda v: 0.08 * (v + 1)

    def weights(v):   # policy version v: mu = atanh-free code (kernels zero -> mu = bmu), log_std = -20 (tanh(-40) = -1)
        out = []
        for k, w in zip(keys, vals):
            w = np.zeros_like(w)
            if "pi" in k and k.endswith("dense_2/bias"):
                w[:] = code(v)
            if "pi" in k and k.endswith("dense_3/bias"):
                w[:] = -40.0
            out.append(w)
        return out
    ps = ddrl.ParameterServer(keys, weights(0))
    rb = ddrl.ReplayBufferSAC1(8, 2, n * steps, seed=0)
    roll = RolloutDevice(ps, rb, opt)
    assert roll._fused_ready()
    push_after = {2: 1, 3: 2, 9: 3, 10: 4, 11: 5, 20: 6, 33: 7}          # step -> version pushed right after it
    EPI = 13                                                          # csrc/env.hip: per-env episode counter
    epi_before = roll.env.get_state()[EPI].cpu().numpy().copy()
    last_end = np.full(n, -1)                                         # the step in which env i last ended an episode (-1: the initial pull)
    pushed_by = lambda s: max([0] + [v for k, v in push_after.items() if k <= s])   # newest version pushed after a step <= s
    for s in range(steps):
        roll.step()
        acts = rb.rings()["acts_buf"][s * n:(s + 1) * n].cpu().numpy()
        used = np.rint(np.arctanh(np.clip(acts[:, 0], -0.999, 0.999)) / 0.08 - 1).astype(int)
        assert (used == used[0]).all() and np.allclose(acts[:, 0], np.tanh(code(used[0])), atol=1e-6)
        # exactly: the push after step k is adopted at the end of step k + 1 and acted on from step k + 2
        assert used[0] == pushed_by(s - 2), (s, used[0], pushed_by(s - 2))
        # never staler than the reference: env i pulled at the end of step last_end[i], seeing pushes after steps < last_end[i]
        ref = np.array([pushed_by(e - 1) if e >= 0 else 0 for e in last_end])
        assert (used >= ref).all(), (s, used[0], ref.max())
        epi_after = roll.env.get_state()[EPI].cpu().numpy()
        last_end[epi_after > epi_before] = s
        epi_before = epi_after.copy()
        if s in push_after:
            ps.push(keys, weights(push_after[s]))
    assert (last_end >= steps - 8).all() and roll.env.stats()[0] >= n * (steps // 7 - 1)    # the episode ends the bound is about did happen


def _coded_weights(keys, vals, v):
    """Policy version v as a recognisable action: zero kernels, mu biases 0.08 (v % 40 + 1) and 0.08 (v // 40 + 1) (two action
    dims = two base-40 digits: tanh keeps 40 levels apart by > 5e-4), log_std at its floor."""
    out = []
    for k, w in zip(keys, vals):
        w = np.zeros_like(w)
        if "pi" in k and k.endswith("dense_2/bias"):
            w[0], w[1:] = 0.08 * (v % 40 + 1), 0.08 * (v // 40 + 1)
        if "pi" in k and k.endswith("dense_3/bias"):
            w[:] = -40.0
        out.append(w)
    return out


def _decode_version(acts):
    d = [np.rint(np.arctanh(np.clip(acts[:, c], -0.9999, 0.9999)) / 0.08 - 1).astype(int) for c in (0, 1)]
    assert np.allclose(acts[:, 0], np.tanh(0.08 * (d[0] + 1)), atol=1e-6) and np.allclose(acts[:, 1], np.tanh(0.08 * (d[1] + 1)), atol=1e-6)
    return d[0] + 40 * d[1]


@pytest.mark.parametrize("n,limit,steps", [(64, 7, 60), (160, 23, 90), (4096, 70, 170), (1024, 300, 330), (10240, 40, 100)])
def test_vectorised_rollout_equals_n_reference_workers_in_the_weights_each_env_acts_on(n, limit, steps):
    """Bar (1) for the vectorised rollout: the reference runs one worker per env and each pulls the server's weights at ITS OWN
    episode end (example/dsac.py:127-130: o = env.reset(); weights = ps.pull(keys); agent.set_weights(keys, weights)) and acts on
    them until its next one.  RolloutDevice(adopt="episode") must act, for every env and every step, on EXACTLY the version that
    env's own worker_rollout would hold: the one the server held when the env's last episode ended (the initial pull before its
    first).  Bias-coded versions make the version behind every stored action readable; pushes land between vector steps, bursts
    of them, long pauses (the single-version fast path), and up to min(n, limit) + 1 versions are live at once.  (10 240 envs: more
    than the planning kernel keeps in registers — eight per thread — so the last 2 048 take its parked-position path.)"""
    import distributed_drl_amd as ddrl
    from distributed_drl_amd.agent import HyperParameters, Learner
    from distributed_drl_amd.workers import RolloutDevice
    opt = HyperParameters()
    opt.num_envs, opt.start_steps, opt.max_ep_len, opt.seed = n, -1, limit, 3
    keys, vals = Learner(opt).get_weights()
    ps = ddrl.ParameterServer(keys, _coded_weights(keys, vals, 0))
    rb = ddrl.ReplayBufferSAC1(8, 2, n * steps, seed=0)
    roll = RolloutDevice(ps, rb, opt)
    assert roll._versions and roll.actor.n_slots == min(n, limit) + 2
    rs = np.random.RandomState(5)
    # version pushed right after step s: a burst at the start, one every step for a while, then nothing for > limit steps, then more
    push_after, v = {}, 0
    for s in range(steps):
        if s < 12 or (20 <= s < 20 + limit + 3) or (s >= 20 + 2 * limit + 12 and rs.rand() < 0.4):
            v += 1
            push_after[s] = v
    EPLEN, EPI = 10, 13                                       # csrc/env.hip: per-env episode length / episode counter
    st0 = roll.env.get_state()
    st0[EPLEN] = torch.arange(n, device="cuda").float() % limit   # stagger the time limits: episode ends in every step, env by env
    roll.env.set_state(st0)
    epi_before = roll.env.get_state()[EPI].cpu().numpy().copy()
    holds = np.zeros(n, int)                                  # the version env i's reference worker holds
    newest, max_live = 0, 0
    for s in range(steps):
        roll.step()
        acts = rb.rings()["acts_buf"][s * n:(s + 1) * n].cpu().numpy()
        np.testing.assert_array_equal(_decode_version(acts), holds, err_msg="step %d" % s)
        epi_after = roll.env.get_state()[EPI].cpu().numpy()
        holds[epi_after > epi_before] = newest               # episode ended in step s: reset, pull what the server holds
        epi_before = epi_after.copy()
        max_live = max(max_live, len(set(holds)))
        if s in push_after:
            newest = push_after[s]
            ps.push(keys, _coded_weights(keys, vals, newest))
    slots, st = roll.actor.version_state()
    assert not st["out_of_slots"] and max_live >= min(limit, 6)
    assert roll.env.stats()[0] >= n * (steps // limit - 1)


@pytest.mark.parametrize("n,limit,steps,slots,hidden", [(96, 9, 30, None, None), (1024, 37, 64, None, None), (1024, 37, 24, 24, None),
                                                        (1024, 37, 24, 100, None), (512, 21, 24, 16, (64, 512)), (256, 11, 16, 40, (128, 96))])
def test_versioned_rollout_actions_match_each_versions_own_policy(n, limit, steps, slots, hidden, monkeypatch):
    """The same with real (random) policies: every env's stored action equals Actor.get_action of THAT env's version on the
    observation it acted on, with the fused step's own noise element — within float32 of the row-major policy kernels.  (The second
    case: 32 row tiles' worth of envs spread over up to 33 live versions — mostly partial row tiles.  `slots`: the forward's planner
    told the chip holds that many workgroups (DDRL_VER_WG_SLOTS), so that these env counts walk what 8192+ envs do on the real
    one — full rounds of coarse workgroups, then the surplus row tiles cut into short ones; `hidden`: other column-tile counts, 16
    of them included — four workgroups per row tile at the least.)"""
    import distributed_drl_amd as ddrl
    from distributed_drl_amd import _lib
    from distributed_drl_amd.agent import Actor, HyperParameters, Learner
    from distributed_drl_amd.workers import RolloutDevice
    if slots is not None:
        monkeypatch.setenv("DDRL_VER_WG_SLOTS", str(slots))
    opt = HyperParameters()
    if hidden is not None:
        opt.hidden_sizes = hidden
    opt.num_envs, opt.start_steps, opt.max_ep_len, opt.seed = n, -1, limit, 11
    keys, vals = Learner(opt).get_weights()
    rs = np.random.RandomState(2)

    def version(v):
        return [(w + (0.3 * rs.standard_normal(w.shape)).astype(np.float32) * (0.2 if w.ndim == 2 else 1.0)) for w in vals]
    vers = {0: version(0)}
    ps = ddrl.ParameterServer(keys, vers[0])
    rb = ddrl.ReplayBufferSAC1(8, 2, n * steps, seed=0)
    roll = RolloutDevice(ps, rb, opt)
    assert roll._versions
    scratch = Actor(opt, max_rows=n, index=99)
    EPLEN, EPI = 10, 13
    st0 = roll.env.get_state()
    st0[EPLEN] = torch.arange(n, device="cuda").float() % limit
    roll.env.set_state(st0)
    epi_before = roll.env.get_state()[EPI].cpu().numpy().copy()
    holds, newest = np.zeros(n, int), 0
    lib = _lib.load()
    for s in range(steps):
        obs = roll.env.obs.clone()
        ctr = roll.actor._noise_ctr
        roll.step()
        eps = torch.empty(n, 2, dtype=torch.float32, device="cuda")
        _lib.check(lib.ddrl_normal_fill(_lib.dptr(eps), n * 2, roll.actor._noise_seed, ctr, _lib.stream_ptr()))
        acts = rb.rings()["acts_buf"][s * n:(s + 1) * n]
        np.testing.assert_array_equal(rb.rings()["obs1_buf"][s * n:(s + 1) * n].cpu().numpy(), obs.cpu().numpy())
        for v in sorted(set(holds)):
            rows = torch.from_numpy(np.nonzero(holds == v)[0]).cuda()
            scratch.set_weights(keys, vers[v])
            want = scratch.get_actions(obs[rows], eps=eps[rows])
            np.testing.assert_allclose(acts[rows].cpu().numpy(), want.cpu().numpy(), rtol=1e-5, atol=1e-5, err_msg="step %d version %d" % (s, v))
        epi_after = roll.env.get_state()[EPI].cpu().numpy()
        holds[epi_after > epi_before] = newest
        epi_before = epi_after.copy()
        if s % 2 == 0:
            newest += 1
            vers[newest] = version(newest)
            ps.push(keys, vers[newest])
    assert len(set(holds)) >= 3


def test_nstep_rollout_weight_adoption_equals_the_reference_workers():
    """The n-step driver's worker (algos/sac1/sac_ray.py:208-262) pulls at ITS episode end, and only once the buffer's `steps` exceed
    start_steps (:259-262).  RolloutDeviceNStep(adopt="episode") must act, per env and per step, on exactly the version that env's own
    worker would hold.  Bias-coded versions, no action noise, staggered episode limits, a push after almost every step."""
    import distributed_drl_amd as ddrl
    from distributed_drl_amd.agent import HyperParameters, Learner
    n, steps = 64, 50
    opt = HyperParameters()
    opt.num_envs, opt.seed, opt.Ln, opt.max_ep_len, opt.action_repeat, opt.start_steps = n, 3, 4, 14, 2, 0
    opt.buffer_size, opt.batch_size, opt.num_buffers = 4000, 8, 1
    opt.obs_noise, opt.act_noise, opt.reward_scale = 0.0, 0.0, 1
    keys, vals = Learner(opt).get_weights()
    ps = ddrl.ParameterServer(keys, _coded_weights(keys, vals, 0))
    rb = ddrl.ReplayBufferNStep(opt)
    ro = ddrl.RolloutDeviceNStep(ps, rb, opt)
    assert ro._versions and ro.limit_steps == 7 and ro.actor.n_slots == 9
    EPLEN = 10
    st0 = ro.env.get_state()
    st0[EPLEN] = torch.arange(n, device="cuda").float() % ro.limit_steps
    ro.env.set_state(st0)
    holds = np.zeros(n, int)       # the version env i's worker holds (the initial pull: version 0)
    newest, learning, seen = 0, False, set()
    for s in range(steps):
        policy_phase = ro.filling_steps > opt.start_steps
        ro.step()
        learning = learning or rb.get_counts()[1] > opt.start_steps       # what the worker reads at its episode end, after its store (:255)
        if policy_phase:
            used = _decode_version(ro.act.cpu().numpy())
            np.testing.assert_array_equal(used, holds, err_msg="step %d" % s)
            seen.update(used.tolist())
        ended = ro.env.ended.cpu().numpy().astype(bool)
        if learning:
            holds[ended] = newest
        if s % 5 != 4:
            newest += 1
            ps.push(keys, _coded_weights(keys, vals, newest))
    assert len(seen) >= 10 and not ro.actor.version_state(with_slots=False)[1]["out_of_slots"]


def test_new_entry_points_refuse_what_they_cannot_do():
    """Error behaviour of the round-4 entry points (every function returns a status, nothing aborts): the version store twice / out of
    range / on a policy outside the direct-operand envelope, versioned get_action without a store, the fused rollout step on a compact
    ring, ddrl_dqn_step_ring on narrow observations."""
    import ctypes
    import distributed_drl_amd as ddrl
    from distributed_drl_amd import _lib, dqn
    from distributed_drl_amd.agent import Actor, HyperParameters
    from distributed_drl_amd.env import VecLunarLander
    lib = _lib.load()
    opt = HyperParameters()
    a = Actor(opt, max_rows=64)
    assert lib.ddrl_actor_act_versioned(a._h, None, None, 64, 1, 10, None, None) == _lib.DDRL_ERR_BAD_ARG            # NULL pointers
    obs = torch.zeros(64, 8, device="cuda")
    act = torch.zeros(64, 2, device="cuda")
    assert lib.ddrl_actor_act_versioned(a._h, _lib.dptr(obs), None, 64, 1, 10, _lib.dptr(act), None) == _lib.DDRL_ERR_BAD_ARG   # no store yet
    assert lib.ddrl_actor_versions_enable(a._h, 1, None) == _lib.DDRL_ERR_BAD_ARG and lib.ddrl_actor_versions_enable(a._h, 5000, None) == _lib.DDRL_ERR_BAD_ARG
    a.enable_versions(4)
    assert lib.ddrl_actor_versions_enable(a._h, 4, None) == _lib.DDRL_ERR_BAD_ARG                                        # already enabled
    assert b"already enabled" in lib.ddrl_last_error()
    assert lib.ddrl_actor_act_versioned(a._h, _lib.dptr(obs), None, 32, 1, 10, _lib.dptr(act), None) == _lib.DDRL_ERR_BAD_ARG   # not all envs
    a.get_actions_versioned(obs, 10, deterministic=True, out=act)
    opt2 = HyperParameters()
    opt2.hidden_sizes = (50, 34)                                                                                         # hidden % 4 != 0: generic kernels only
    with pytest.raises(ValueError, match="direct-operand"):
        Actor(opt2, max_rows=64).enable_versions(4)
    # a store that runs out of slots says so (two slots, three versions in use) instead of corrupting a version some env acts on silently
    b = Actor(opt, max_rows=64)
    b.enable_versions(2)
    keys, vals = b.get_weights()
    ended = torch.zeros(64, dtype=torch.uint8, device="cuda")
    b.set_weights(keys, vals)                      # version 1 (slot 1); nobody on it yet
    ended[:10] = 1
    b.adopt_where_ended(ended)                     # ten envs move to slot 1: both slots in use
    b.set_weights(keys, vals)                      # a third version: no free slot
    assert b.version_state(with_slots=False)[1]["out_of_slots"]
    # the fused rollout step stores float32 rows: a compact ring is refused
    env = VecLunarLander(64, seed=0)
    cring = ddrl.ReplayBuffer(8, 2, 256, compact_obs=True)
    assert lib.ddrl_rollout_begin(env._h, a._h, None) == 0
    assert lib.ddrl_rollout_step(env._h, a._h, cring._h, 1, 0, 0, 1, None, None, None) == _lib.DDRL_ERR_BAD_ARG
    assert b"float32 rings only" in lib.ddrl_last_error()

    class O:
        obs_dim, act_dim, hidden_size, gamma, lr, polyak, batch_size, seed, buffer_size, save_dir = 12, 3, [16, 8], 0.99, 1e-3, 0.995, 8, 0, 64, "."
    ln = dqn.Learner(O, "learner")
    ring = ddrl.ReplayBufferDQN(O, 0, seed=1)
    assert lib.ddrl_dqn_step_ring(ln._h, ring._h, None, None, None, None) == _lib.DDRL_ERR_UNSUPPORTED
    with pytest.raises(ValueError, match="high <= 0"):
        ln.train_from(ring)                        # the fallback's sample_batch on an empty ring: the reference's ValueError


@pytest.mark.parametrize("variant", ["ddqn", "sqn"])
def test_dqn_shaped_driver_end_to_end(tmp_path, variant):
    """algos/dqn/train.py's __main__ (374-462) in small: a node's parameter server and two buffers, rollout workers storing into random
    buffers, the learner fed by the Cache helper, the tester's round (TensorBoard scalars, weight pickle, checkpoint of server and
    buffers) — real classes on the GPU, a scripted host env in place of TradingEnv.  Also algos/sqn/train.py (the same driver)."""
    import pickle
    import threading
    import distributed_drl_amd as ddrl
    from distributed_drl_amd import dqn, workers
    from distributed_drl_amd.logx import read_scalars

    class Opt:
        pass
    opt = Opt()
    opt.obs_dim, opt.act_dim, opt.hidden_size, opt.gamma, opt.lr, opt.polyak, opt.batch_size, opt.seed, opt.alpha = 6, 3, [32, 24], 0.99, 1e-3, 0.995, 16, 1, 0.1
    opt.buffer_size, opt.num_buffers, opt.num_nodes, opt.start_steps, opt.recover, opt.push_freq = 500, 2, 1, 40, False, 5
    opt.save_dir, opt.summary_dir, opt.save_interval, opt.checkpoint_freq = str(tmp_path), str(tmp_path / "tb"), 10, 1e-3
    opt.env_name, opt.exp_name, opt.num_workers, opt.a_l_ratio, opt.max_updates = "Scripted", "t", 2, 10, 25
    L, A = (dqn.LearnerSQN, dqn.ActorSQN) if variant == "sqn" else (dqn.Learner, dqn.Actor)

    class Env:
        """Episodes of 7 steps; reward = 1 when the action equals the step's parity class."""

        class _Space:
            def __init__(self, rs):
                self.rs = rs

            def sample(self):
                return int(self.rs.randint(0, 3))

        def __init__(self, seed=0):
            self.rs = np.random.RandomState(seed)
            self.action_space = Env._Space(self.rs)
            self.rewards = [0.0]

        def reset(self):
            self.k, self.rewards = 0, [0.0]
            return self.rs.rand(6)

        def step(self, a):
            self.k += 1
            r = 1.0 if int(a) == self.k % 3 else 0.0
            self.rewards[0] += r
            return self.rs.rand(6), r, self.k >= 7, {}

    keys, values = L(opt, "ps").get_weights()
    ps = ddrl.ParameterServerNode(opt, keys=keys, values=values)
    bufs = [ddrl.ReplayBufferDQN(opt, i, seed=i) for i in range(opt.num_buffers)]
    node_buffer = [bufs]
    # ---- rollouts: two workers, stopped once the buffers hold enough
    opt.stop_event = threading.Event()
    threads = [threading.Thread(target=workers.worker_rollout_dqn, args=(ps, bufs, opt, i),
                                kwargs=dict(make_env=lambda i=i: Env(i), make_agent=lambda o: A(o, "worker")), daemon=True) for i in range(2)]
    for t in threads:
        t.start()
    t0 = time.time()
    while sum(b.get_counts()[1] for b in bufs) < 150 and time.time() - t0 < 60:
        time.sleep(0.01)
    opt.stop_event.set()
    for t in threads:
        t.join(30)
    opt.stop_event = None
    _, actor_steps, sizes = workers.get_al_status(node_buffer, opt)[0], [b.get_counts()[1] for b in bufs], [b.get_counts()[2] for b in bufs]
    assert sum(actor_steps) >= 150 and min(actor_steps) > 20          # both buffers were stored into (a random one per transition)
    # ---- learner: 25 updates through the Cache helper, weights to the server every 5
    v0 = ps.get_weights()[keys[0]].copy()
    n = workers.worker_train_dqn(ps, node_buffer, opt, 0, node_ps=[ps], make_agent=lambda o: L(o, "learner"))
    assert n == 25 and ps.learner_step == 25 and np.abs(ps.get_weights()[keys[0]] - v0).max() > 0
    assert sum(b.get_counts()[0] for b in bufs) >= 25                   # sample_batch counts as the reference's learner_steps
    # ---- the device-resident learner (no Cache): the same schedule of buffers as the Cache's draws, every update out of that buffer's sampler
    class Rng:
        def __init__(self):
            self.rs, self.picks = np.random.RandomState(4), []

        def choice(self, n, k):
            v = self.rs.choice(n, k)
            self.picks.append(int(v[0]))
            return v
    before = [b.get_counts()[0] for b in bufs]
    rng = Rng()
    td = workers.TrainDeviceDQN([ps], node_buffer, opt, make_agent=lambda o: L(o, "learner"), rng=rng)
    step0 = ps.learner_step
    assert td.run(10) == 10 and ps.learner_step == step0 + 10
    drawn = [b.get_counts()[0] - x for b, x in zip(bufs, before)]
    assert drawn == [rng.picks[1::2].count(0), rng.picks[1::2].count(1)] and sum(drawn) == 10
    np.testing.assert_array_equal(ps.get_weights()[keys[0]], td.agent.get_weights()[1][0])    # the push after update 10 is what the server holds
    # ---- tester: one round
    ret = workers.worker_test_dqn(ps, node_buffer, opt, node_ps=[ps], make_env=lambda: Env(9), make_agent=lambda o: A(o, "test"), log=lambda s: None,
                                  wait=lambda ops, num_returns: None, max_rounds=1)
    assert 0.0 <= ret <= 7.0
    saved = [f for f in os.listdir(tmp_path) if f.endswith("_weights.pickle")]
    assert len(saved) == 1 and sorted(pickle.load(open(tmp_path / saved[0], "rb"))) == sorted(ps.get_weights())
    assert os.path.exists(tmp_path / "checkpoint" / "checkpoint_weights.pickle") and os.path.exists(tmp_path / "checkpoint" / "obs1_buf-1.npy")
    runs = os.listdir(tmp_path / "tb")
    assert len(runs) == 1 and "Scripted-t-workers_num:2%10" in runs[0]
    run = tmp_path / "tb" / runs[0]
    sc = read_scalars(str(run / os.listdir(run)[0]))
    assert sorted(t for _, t, _ in sc) == ["Reward", "a_l_ratio", "score", "update_frequency"]
    assert [v for _, t, v in sc if t == "Reward"][0] == pytest.approx(ret) and {st for st, _, _ in sc} == {sum(b.get_counts()[0] for b in bufs)}


def test_reference_style_nstep_rollout_stores_the_deque_windows():
    """worker_rollout_nstep (algos/sac1/sac_ray.py:178-274) with the real Actor, parameter server, host lander behind the host
    `Wrapper`, and a real n-step ring: every window in the ring is what the worker's two deques held at that store — rebuilt here
    from a recording of the env's steps — and the counters advance as the reference's (num_buffers per store)."""
    from collections import deque
    import distributed_drl_amd as d
    from distributed_drl_amd import workers
    from distributed_drl_amd.agent import Actor, HyperParameters
    from distributed_drl_amd.env import LunarLander, Wrapper
    opt = HyperParameters()
    opt.seed, opt.Ln, opt.max_ep_len, opt.action_repeat, opt.save_freq, opt.start_steps = 3, 4, 30, 2, 1, 12
    opt.buffer_size, opt.batch_size, opt.num_buffers, opt.weights_file = 400, 8, 1, ""
    opt.obs_noise, opt.act_noise, opt.reward_scale = 0.01, 0.2, 5
    rb = d.ReplayBufferNStep(opt)
    ps = d.ParameterServer(*Actor(opt, job="worker").get_weights())
    log = []

    class Rec:
        """The wrapped env as the worker sees it, with every reset / step recorded."""

        def __init__(self):
            self._w = Wrapper(LunarLander(seed=5, max_ep_len=1000), opt.obs_noise, opt.act_noise, opt.reward_scale, 3, rng=np.random.RandomState(1))
            self.action_space = self._w.action_space
            self.n = 0

        def reset(self):
            o = self._w.reset()
            log.append(("reset", o.copy()))
            return o

        def step(self, a):
            o2, r, dd, info = self._w.step(a)
            log.append(("step", np.array(a, np.float32).copy(), float(r), bool(dd), o2.copy()))   # `a` AFTER the in-place noise: what is queued
            self.n += 1
            if self.n >= 90:
                opt.stop_event.set()
            return o2, r, dd, info

    import threading
    opt.stop_event = threading.Event()
    workers.worker_rollout_nstep(ps, [rb], opt, 0, make_env=Rec)
    # ---- the reference's loop over the recording
    oq, aq, want, tq, ep_len = deque([], maxlen=opt.Ln + 1), deque([], maxlen=opt.Ln), [], 1, 0
    for e in log:
        if e[0] == "reset":
            oq.append(e[1]); tq, ep_len = 1, 0
            continue
        _, a, r, dd, o2 = e
        aq.append((a, r, dd)); oq.append(o2)
        if tq >= opt.Ln and tq % opt.save_freq == 0:
            want.append((np.stack(list(oq)), np.stack([x[0] for x in aq]), np.array([x[1] for x in aq]), np.array([x[2] for x in aq], np.float32)))
        tq += 1
        ep_len += 1
    assert len(want) > 40 and sum(1 for e in log if e[0] == "reset") >= 3
    rings = rb.rings()
    for k, j in (("buffer_o", 0), ("buffer_a", 1), ("buffer_r", 2), ("buffer_d", 3)):
        got = rings[k][:len(want)].cpu().numpy()
        np.testing.assert_array_equal(got, np.stack([w[j] for w in want]).astype(np.float32).reshape(got.shape), err_msg=k)
    assert rb.get_counts() == (0, len(want) * opt.num_buffers, len(want))


@pytest.mark.parametrize("hidden,obs,act", [((400, 300), 8, 2), ((64, 512), 11, 1), ((130, 70), 36, 4), ((512, 36), 3, 3), ((7, 9), 1, 1)])
def test_get_action_as_one_launch_equals_the_batched_kernels(hidden, obs, act):
    """Actor.get_action(o) (actor_learner.py:195-197) runs as ONE launch (ddrl_actor_act_one: both layers, head, squash, noise from the
    counter): the same action as get_actions on the row with the noise elements ddrl_normal_fill yields at the same stream position —
    within float32 summation order — stochastic and deterministic, across a set_weights, and the noise counter advances as before."""
    from distributed_drl_amd import _lib
    from distributed_drl_amd.agent import Actor, HyperParameters
    opt = HyperParameters()
    opt.hidden_sizes, opt.obs_dim, opt.act_dim, opt.seed = hidden, obs, act, 5
    a = Actor(opt, max_rows=32)
    lib = _lib.load()
    rs = np.random.RandomState(1)
    for rnd in range(3):
        if rnd == 1:
            a._flat_set(torch.from_numpy((0.3 * rs.standard_normal(a.get_weights_flat().numel())).astype(np.float32)).cuda())
        for det in (False, True):
            o = rs.randn(obs).astype(np.float32) * 2
            ctr = a._noise_ctr
            got = a.get_action(o, deterministic=det)
            assert a._noise_ctr == ctr + (0 if det else act)
            eps = torch.zeros(1, act, dtype=torch.float32, device="cuda")
            _lib.check(lib.ddrl_normal_fill(_lib.dptr(eps), act, a._noise_seed, ctr, _lib.stream_ptr()))
            want = a.get_actions(torch.from_numpy(o).cuda().reshape(1, -1), deterministic=det, eps=eps)[0].cpu().numpy()
            np.testing.assert_allclose(got, want, rtol=2e-5, atol=2e-6, err_msg="round %d det %s" % (rnd, det))
    assert getattr(a, "_act_one", True) is True
