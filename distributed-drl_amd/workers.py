"""worker_rollout / worker_train / worker_test with the reference's call shapes.

Mirrors:
  example/dsac.py:76-130    worker_rollout(ps, replay_buffer, args)
  example/dsac.py:133-150   worker_train(ps, replay_buffer, args)
  example/dsac.py:153-177   worker_test(ps, start_time)            (here: explicit args)
  algos/sac1/sac1.py:133-154,157-213,216-252   the SAC1 flavours (opt, index; a_l_ratio throttle)
  algos/dqn/train.py:177-371 (= algos/sqn/train.py)   the DQN / SQN driver: Cache, worker_train, worker_rollout, worker_test, get_al_status
  algos/sac1/sac_ray.py:123-274                        the n-step driver: Cache (single server), worker_train, worker_rollout (window deques)

`ps` / `replay_buffer` are actor handles (`remote.py` shim or Ray): methods are invoked as
`handle.method.remote(...)` and awaited with `get`.  Two execution styles share these semantics:

  * reference style (one env per worker, one transition per store RPC) — `worker_rollout`,
    `worker_train`: exactly the reference's event order (pinned by tests/test_workers_cpu.py
    against traces recorded from the reference's own functions);
  * device style — `rollout_loop_device`, `train_loop_device`: `opt.num_envs` environments per
    worker stepped by one kernel, `store_batch` of num_envs transitions, batched policy forward,
    learner sampling straight out of HBM.  Per-env semantics are those of num_envs independent
    reference workers, except that fresh weights are adopted at the next vector step after a
    push (the reference adopts them at each worker's episode end) — never staler than the
    reference.
"""
import time

from . import remote as _ray


def _remote(method, *args):
    """handle.method.remote(*args) for actor handles; direct call for plain objects."""
    r = getattr(method, "remote", None)
    return r(*args) if r is not None else method(*args)


def _get(x):
    return _ray.get(x)


def _default_env(name, args):
    from . import env as _env
    return _env.make(name, seed=int(getattr(args, "seed", 0)), max_ep_len=int(args.max_ep_len))


def _stop(args):
    ev = getattr(args, "stop_event", None)
    return ev is not None and ev.is_set()


# ------------------------------------------------------------------------------------------
# example/dsac.py flavour
# ------------------------------------------------------------------------------------------
def worker_rollout(ps, replay_buffer, args, make_env=None, make_agent=None):
    """example/dsac.py:76-130."""
    if make_env is None:
        make_env = lambda name: _default_env(name, args)
    if make_agent is None:
        from .agent import Actor, Model
        # with example/dsac.py's own `args` the rollout agent is a whole Model (dsac.py:83), whose "main"
        # variables include q1, q2, v: pull(keys) must ask for the same key list the learner pushes
        make_agent = (lambda a: Model(a)) if hasattr(args, "ac_kwargs") else (lambda a: Actor(a, job="worker"))
    env = make_env(args.env)
    o, r, d, ep_ret, ep_len = env.reset(), 0, False, 0, 0
    total_steps = args.steps_per_epoch * args.epochs

    agent = make_agent(args)
    keys = agent.get_weights()[0]
    weights = _get(_remote(ps.pull, keys))
    agent.set_weights(keys, weights)

    for t in range(total_steps):
        if _stop(args):
            break
        # uniform-random actions for the first start_steps+1 steps (strict '>': dsac.py:96)
        if t > args.start_steps:
            a = agent.get_action(o)
        else:
            a = env.action_space.sample()
        o2, r, d, _ = env.step(a)
        ep_ret += r
        ep_len += 1
        # hitting the time horizon is not a terminal state (dsac.py:109)
        d = False if ep_len == args.max_ep_len else d
        _remote(replay_buffer.store, o, a, r, o2, d)  # fire-and-forget (dsac.py:112)
        o = o2
        if d or (ep_len == args.max_ep_len):
            o, r, d, ep_ret, ep_len = env.reset(), 0, False, 0, 0
            weights = _get(_remote(ps.pull, keys))
            agent.set_weights(keys, weights)


def worker_train(ps, replay_buffer, args, make_agent=None):
    """example/dsac.py:133-150: pull, then `train; push every 300th update` forever."""
    if make_agent is None:
        from .agent import Learner, Model
        if hasattr(args, "ac_kwargs"):
            # example/dsac.py's own `args` (dsac.py:185-216): its algorithm is the SAC-v of example/model.py
            make_agent = lambda a: Model(a)
        else:
            class _Model(Learner):  # example/model.py:92-101's surface over the SAC1 learner: train = sample_batch RPC + one step
                def train(self, replay_buffer, args):
                    batch = _get(_remote(replay_buffer.sample_batch, args.batch_size))
                    return super().train(batch)
            make_agent = lambda a: _Model(a, job="learner")
    agent = make_agent(args)
    keys = agent.get_weights()[0]
    weights = _get(_remote(ps.pull, keys))
    agent.set_weights(keys, weights)

    push_freq = int(getattr(args, "push_freq", 300))
    max_updates = getattr(args, "max_updates", None)
    cnt = 1
    while True:
        agent.train(replay_buffer, args)
        if cnt % push_freq == 0:
            keys, values = agent.get_weights()
            _remote(ps.push, keys, values)
        if (max_updates is not None and cnt >= max_updates) or _stop(args):
            return cnt
        cnt += 1


def worker_test(ps, args, start_time=None, n=10, make_env=None, make_agent=None, log=print, max_rounds=None):
    """example/dsac.py:153-177: pull, run n deterministic episodes, log, repeat.  With `args.logger_kwargs`
    (dsac.py:205) the rounds also go to an EpochLogger: config.json + progress.txt (AverageTestEpRet, Time)."""
    logger = None
    if getattr(args, "logger_kwargs", None):
        from .logx import EpochLogger
        logger = EpochLogger(**args.logger_kwargs)
        logger.save_config({k: v for k, v in vars(args).items()} if hasattr(args, "__dict__") else {})
    if make_env is None:
        make_env = lambda name: _default_env(name, args)
    if make_agent is None:
        from .agent import Actor
        make_agent = lambda a: Actor(a, job="test")   # example/dsac.py's test worker logs through EpochLogger only (no tf.summary writer)
    start_time = time.time() if start_time is None else start_time
    agent = make_agent(args)
    keys = agent.get_weights()[0]
    test_env = make_env(args.env)
    rounds, last = 0, None
    while True:
        weights = _get(_remote(ps.pull, keys))
        agent.set_weights(keys, weights)
        last = agent.test(test_env, None, n)
        log("AverageTestEpRet %.3f  Time %.1f" % (last, time.time() - start_time))
        if logger is not None:
            logger.log_tabular('AverageTestEpRet', last)
            logger.log_tabular('Time', time.time() - start_time)
            logger.dump_tabular()
        rounds += 1
        if (max_rounds is not None and rounds >= max_rounds) or _stop(args):
            return last


# ------------------------------------------------------------------------------------------
# algos/sac1/sac1.py flavour (opt, index; throttle on steps / sample_times)
# ------------------------------------------------------------------------------------------
def worker_rollout_sac1(ps, replay_buffer, opt, worker_index, make_env=None, make_agent=None, sleep=time.sleep):
    """algos/sac1/sac1.py:157-213."""
    if make_env is None:
        make_env = lambda name: _default_env(name, opt)
    if make_agent is None:
        from .agent import Actor
        make_agent = lambda o_: Actor(o_, job="worker")
    env = make_env(opt.env_name)
    agent = make_agent(opt)
    keys = agent.get_weights()[0]
    o, r, d, ep_ret, ep_len = env.reset(), 0, False, 0, 0
    weights = _get(_remote(ps.pull, keys))
    agent.set_weights(keys, weights)

    t = 0
    while not _stop(opt):
        if t > opt.start_steps:
            a = agent.get_action(o)
        else:
            a = env.action_space.sample()
        t += 1
        o2, r, d, _ = env.step(a)
        ep_ret += r
        ep_len += 1
        d = False if ep_len == opt.max_ep_len else d
        _remote(replay_buffer.store, o, a, r, o2, d)
        o = o2
        if d or (ep_len == opt.max_ep_len):
            # actor/learner ratio gate (sac1.py:203-207)
            sample_times, steps, _ = _get(_remote(replay_buffer.get_counts))
            while sample_times > 0 and steps / sample_times > opt.a_l_ratio:
                sample_times, steps, _ = _get(_remote(replay_buffer.get_counts))
                sleep(0.1)
            weights = _get(_remote(ps.pull, keys))
            agent.set_weights(keys, weights)
            o, r, d, ep_ret, ep_len = env.reset(), 0, False, 0, 0


def worker_train_sac1(ps, replay_buffer, opt, learner_index, make_agent=None, make_cache=None):
    """algos/sac1/sac1.py:133-154: the learner behind the `Cache` helper of sac1.py:103-130 — up to ten sampled batches wait in
    q1 (the helper keeps calling `replay_buffer.sample_batch(opt.batch_size)`), the learner takes them in order, and every
    push_freq-th update's weights go through q2 to `ps.push`.  On this host-buffer surface a batch crosses PCIe down and up again;
    the helper's sample of batch i + 1 runs — on the replay actor's own stream (remote.py) — beside the learner's update i, which is
    what the reference's Cache is for.  The batches and their order are exactly those of the loop without the helper
    (`batch = sample_batch(); train(batch)`): one buffer, one FIFO queue (tests/test_gpu_driver.py).
    make_cache=False: no helper (rounds 1-5).  make_cache="prefetch": the Cache inside the buffer (ReplayBuffer.prefetch: ten draws
    always in flight on the buffer's stream, no helper thread — three Python threads share one interpreter lock here, where the
    reference's helper is a process), then the plain loop."""
    if make_agent is None:
        from .agent import Learner
        make_agent = lambda o_: Learner(o_, job="learner")
    if make_cache == "prefetch":
        _get(_remote(replay_buffer.prefetch, opt.batch_size))
        make_cache = False
    if make_cache is None:
        make_cache = lambda rb: BatchCache(rb, opt, [ps], nodes=None)
    agent = make_agent(opt)
    keys = agent.get_weights()[0]
    weights = _get(_remote(ps.pull, keys))
    agent.set_weights(keys, weights)
    push_freq = int(getattr(opt, "push_freq", 300))
    max_updates = getattr(opt, "max_updates", None)
    cache = make_cache(replay_buffer) if make_cache else None
    if cache is not None:
        cache.start()
    cnt = 1
    while True:
        batch = cache.q1.get() if cache is not None else _get(_remote(replay_buffer.sample_batch, opt.batch_size))
        agent.train(batch)
        if cnt % push_freq == 0:
            if cache is not None:
                cache.q2.put(agent.get_weights())
            else:
                keys, values = agent.get_weights()
                _remote(ps.push, keys, values)
        if (max_updates is not None and cnt >= max_updates) or _stop(opt):
            end = getattr(cache, "end", None)
            if end is not None:
                end()
            return cnt
        cnt += 1


def worker_test_sac1(ps, replay_buffer, opt, make_env=None, make_agent=None, log=print, sleep=time.sleep, max_rounds=None, n=25):
    """algos/sac1/sac1.py:214-252: pull, 25 deterministic episodes, print test_reward / counters / update frequency,
    `ps.save_weights()` whenever the return beats the best so far, sleep 5 s, repeat."""
    if make_env is None:
        make_env = lambda name: _default_env(name, opt)
    if make_agent is None:
        from .agent import Actor
        make_agent = lambda o: Actor(o, job="main")
    agent = make_agent(opt)
    keys, weights = agent.get_weights()
    time0 = time1 = time.time()
    sample_times1, steps, size = _get(_remote(replay_buffer.get_counts))
    max_ret = -1000
    env = make_env(opt.env_name)
    rounds = 0
    while True:
        weights = _get(_remote(ps.pull, keys))
        agent.set_weights(keys, weights)
        ep_ret = agent.test(env, replay_buffer, n)
        sample_times2, steps, size = _get(_remote(replay_buffer.get_counts))
        time2 = time.time()
        log("test_reward: %s sample_times: %s steps: %s buffer_size: %s" % (ep_ret, sample_times2, steps, size))
        log("update frequency: %s total time: %s" % ((sample_times2 - sample_times1) / max(time2 - time1, 1e-9), time2 - time0))
        if ep_ret > max_ret:
            _remote(ps.save_weights, getattr(opt, "save_dir", "") and (opt.save_dir + "/") or "")
            log("****** weights saved! ******")
            max_ret = ep_ret
        time1, sample_times1 = time2, sample_times2
        rounds += 1
        if (max_rounds is not None and rounds >= max_rounds) or _stop(opt):
            return max_ret
        sleep(5)


# ------------------------------------------------------------------------------------------
# algos/dqn/train.py flavour (algos/sqn/train.py is the same driver around the soft-Q agent): node_buffer[node][buffer] handles,
# a random buffer per store, the learner fed by a Cache helper.  The reference reads `opt` and `node_ps` as module globals; here
# they are arguments.
# ------------------------------------------------------------------------------------------
def get_al_status(node_buffer, opt):
    """algos/dqn/train.py:358-371: (actor_steps, learner_steps, cur_size) of every buffer of every node, as arrays."""
    import numpy as np
    learner, actor, size = [], [], []
    for node_index in range(opt.num_nodes):
        for i in range(opt.num_buffers):
            learner_step, actor_step, cur_size = _get(_remote(node_buffer[node_index][i].get_counts))
            learner.append(learner_step)
            actor.append(actor_step)
            size.append(cur_size)
    return np.array(actor), np.array(learner), np.array(size)


class BatchCache:
    """algos/dqn/train.py:177-210 `Cache`: a helper that keeps up to ten sampled batches waiting in q1 (each from a random buffer of a
    random node) and forwards the weights the learner leaves in q2 to EVERY node's parameter server.  The reference runs it as a
    daemon process; device-backed handles cannot cross a process boundary, so this one is a daemon thread over the same two queues."""

    def __init__(self, node_buffer, opt, node_ps, rng=None, nodes=True):
        """nodes=False: the single-server form of algos/sac1/sac_ray.py:123-153 — `node_buffer` is one flat list of buffers, the
        weights go to node_ps[0], q1 holds ten.  nodes=None: algos/sac1/sac1.py:103-130 — `node_buffer` is THE replay buffer (one
        handle), every batch is `sample_batch(opt.batch_size)` of it, the weights go to node_ps[0], q1 holds ten."""
        import queue
        import threading
        self.node_buffer, self.opt, self.node_ps, self.nodes = node_buffer, opt, node_ps, nodes
        self.rng = rng
        self._stop, self._wake = threading.Event(), threading.Event()
        wake = self._wake

        class _Q(queue.Queue):
            """queue.Queue whose get / put wake the helper: the reference's helper is a PROCESS that spins on qsize() / empty()
            (algos/dqn/train.py:195-203); a thread doing that holds the GIL against the learner, so this one sleeps until a queue moves."""

            def get(self, block=True, timeout=None):
                item = queue.Queue.get(self, block, timeout)
                wake.set()
                return item

            def put(self, item, block=True, timeout=None):
                queue.Queue.put(self, item, block, timeout)
                wake.set()

        self.q1, self.q2 = _Q(12 if nodes else 10), _Q(5)
        self.p1 = threading.Thread(target=self.ps_update, args=(self.q1, self.q2, self.node_buffer), daemon=True)

    def _one_batch(self, q1, node_buffer):
        import numpy as np
        if self.nodes is None:
            q1.put(_get(_remote(node_buffer.sample_batch, self.opt.batch_size)))   # (fresh arrays per call: the reference's deepcopy)
            return
        rng = self.rng if self.rng is not None else np.random
        if not self.nodes:
            q1.put(_get(_remote(node_buffer[rng.choice(self.opt.num_buffers, 1)[0]].sample_batch)))
            return
        node_idx = rng.choice(self.opt.num_nodes, 1)[0]
        buffer_idx = rng.choice(self.opt.num_buffers, 1)[0]
        q1.put(_get(_remote(node_buffer[node_idx][buffer_idx].sample_batch)))   # (fresh arrays per call: the reference's deepcopy)

    def ps_update(self, q1, q2, node_buffer):
        self._one_batch(q1, node_buffer)
        while not self._stop.is_set():
            self._wake.clear()
            idle = True
            if q1.qsize() < 10:
                self._one_batch(q1, node_buffer)
                idle = False
            if not q2.empty():
                self._forward(q2)
                idle = False
            if idle:                 # ten batches waiting, no weights to forward: sleep until the learner takes a batch or leaves weights
                self._wake.wait(0.05)
        while not q2.empty():        # end(): the weights still waiting go out (the reference terminates its helper process: they are lost there)
            self._forward(q2)

    def _forward(self, q2):
        keys, values = q2.get()
        for i in range(self.opt.num_nodes if self.nodes else 1):
            _remote(self.node_ps[i].push, keys, values)

    def start(self):
        self.p1.start()

    def end(self):
        self._stop.set()
        self._wake.set()
        while True:                  # a helper blocked on a full q1 is let through
            try:
                self.q1.get_nowait()
            except Exception:
                break
        self.p1.join(5)


def worker_train_dqn(ps, node_buffer, opt, learner_index, node_ps=None, make_agent=None, make_cache=None):
    """algos/dqn/train.py:213-231."""
    if make_agent is None:
        from .dqn import Learner
        make_agent = lambda o_: Learner(o_, job="learner")
    if make_cache is None:
        make_cache = lambda nb: BatchCache(nb, opt, node_ps if node_ps is not None else [ps])
    agent = make_agent(opt)
    keys = agent.get_weights()[0]
    weights = _get(_remote(ps.pull, keys))
    agent.set_weights(keys, weights)
    cache = make_cache(node_buffer)
    cache.start()
    max_updates = getattr(opt, "max_updates", None)
    cnt = 1
    while True:
        batch = cache.q1.get()
        agent.train(batch, cnt)
        if cnt % opt.push_freq == 0:
            cache.q2.put(agent.get_weights())
        if (max_updates is not None and cnt >= max_updates) or _stop(opt):
            end = getattr(cache, "end", None)
            if end is not None:
                end()
            return cnt
        cnt += 1


def worker_rollout_dqn(ps, replay_buffer, opt, worker_index, make_env=None, make_agent=None, rng=None):
    """algos/dqn/train.py:234-287.  `replay_buffer` is the node's list of buffers: the counters that decide between policy and random
    actions are read from ONE random buffer at each episode start (and scaled by num_buffers), every transition goes to a random
    buffer; an episode ends at the env's own terminal only.  `rng`: seed() / choice(n, 1) (default np.random — the reference
    reseeds from the OS before every choice)."""
    import numpy as np
    if rng is None:
        rng = np.random
    if make_agent is None:
        from .dqn import Actor
        make_agent = lambda o_: Actor(o_, job="worker")
    if make_env is None:
        make_env = lambda: _default_env(opt.env_name, opt)
    agent = make_agent(opt)
    keys = agent.get_weights()[0]
    rng.seed()
    env = make_env()
    while not _stop(opt):
        o, r, d, ep_ret, ep_len = env.reset(), 0, False, 0, 0
        weights = _get(_remote(ps.pull, keys))
        agent.set_weights(keys, weights)
        rng.seed()
        rand_buff = rng.choice(opt.num_buffers, 1)[0]
        last_learner_steps, last_actor_steps, _size = _get(_remote(replay_buffer[rand_buff].get_counts))
        while True:
            if last_actor_steps * opt.num_buffers > opt.start_steps or opt.recover:
                a = agent.get_action(o)
            else:
                a = env.action_space.sample()
            o2, r, d, _ = env.step(a)
            ep_ret += r
            ep_len += 1
            rng.seed()
            rand_buff = rng.choice(opt.num_buffers, 1)[0]
            _remote(replay_buffer[rand_buff].store, o, a, r, o2, d, worker_index)
            o = o2
            if d:
                break


def worker_test_dqn(ps, node_buffer, opt, node_ps=None, make_env=None, make_agent=None, clock=time.time, log=print, wait=None,
                    max_rounds=None):
    """algos/dqn/train.py:289-355: pull everything, 10 test episodes, actor / learner counters before and after (a_l_ratio and the
    learner's update frequency over the test's duration), TensorBoard scalars, the whole weight dict pickled every save_interval
    learner steps, every server and every buffer checkpointed every checkpoint_freq seconds."""
    import pickle
    import numpy as np
    if make_agent is None:
        from .dqn import Actor
        make_agent = lambda o_: Actor(o_, job="test")
    if make_env is None:
        make_env = lambda: _default_env(opt.env_name, opt)
    if node_ps is None:
        node_ps = [ps]
    if wait is None:
        wait = lambda ops, num_returns: _ray.wait(ops, num_returns=num_returns)
    agent = make_agent(opt)
    keys = agent.get_weights()[0]
    test_env = make_env()
    init_time = clock()
    save_times = checkpoint_times = rounds = 0
    while True:
        weights_all = _get(_remote(ps.get_weights))     # all of them: saved to disk below
        agent.set_weights(keys, [weights_all[key] for key in keys])
        start_actor_step, start_learner_step, _ = get_al_status(node_buffer, opt)
        start_time = clock()
        ave_test_reward, ave_score = agent.test(test_env, 10)
        last_actor_step, last_learner_step, _ = get_al_status(node_buffer, opt)
        actor_step = np.sum(last_actor_step) - np.sum(start_actor_step)
        learner_step = np.sum(last_learner_step) - np.sum(start_learner_step)
        alratio = actor_step / (learner_step + 1)
        update_frequency = int(learner_step / (clock() - start_time))
        total_learner_step = np.sum(last_learner_step)
        log("---------------------------------------------------")
        log("average test reward: %s" % ave_test_reward)
        log("average test score: %s" % ave_score)
        log("frame freq: %s" % np.round((last_actor_step - start_actor_step) / (clock() - start_time)))
        log("actor_steps: %s learner_step: %s" % (np.sum(last_actor_step), total_learner_step))
        log("actor leaner ratio: %.2f" % alratio)
        log("learner freq: %s" % update_frequency)
        log("---------------------------------------------------")
        if learner_step < 100:
            alratio = 0
        agent.write_tb(ave_test_reward, ave_score, alratio, update_frequency, total_learner_step)
        total_time = clock() - init_time
        if total_learner_step // opt.save_interval > save_times:
            with open(opt.save_dir + "/" + str(total_learner_step / 1e6) + "M_" + str(ave_test_reward) + "_weights.pickle", "wb") as pickle_out:
                pickle.dump(weights_all, pickle_out)
                log("****** Weights saved by time! ******")
            save_times = total_learner_step // opt.save_interval
        if total_time // opt.checkpoint_freq > checkpoint_times:
            log("save everything!")
            save_start_time = clock()
            ps_save_op = [_remote(node_ps[i].save_weights) for i in range(opt.num_nodes)]
            buffer_save_op = [_remote(node_buffer[node_index][i].save) for i in range(opt.num_buffers) for node_index in range(opt.num_nodes)]
            wait(buffer_save_op + ps_save_op, opt.num_nodes * opt.num_buffers + 1)
            log("total time for saving : %s" % (clock() - save_start_time))
            checkpoint_times = total_time // opt.checkpoint_freq
        rounds += 1
        if (max_rounds is not None and rounds >= max_rounds) or _stop(opt):
            return ave_test_reward


# ------------------------------------------------------------------------------------------
# algos/sac1/sac_ray.py flavour: the n-step driver (windows of Ln + 1 observations / Ln (a, r, d) triples out of per-worker deques)
# ------------------------------------------------------------------------------------------
def worker_rollout_nstep(ps, replay_buffer, opt, worker_index, make_env=None, make_agent=None, rng=None):
    """algos/sac1/sac_ray.py:178-274 (opt.model != "cnn").  Random actions until `filling_steps` (counted on random actions only)
    exceeds start_steps, or at once with opt.weights_file; a window goes to a random buffer whenever t_queue >= Ln and t_queue %
    save_freq == 0; an episode ends at d or ep_len * action_repeat >= max_ep_len; the worker pulls at an episode end only once
    buffer 0's `steps` exceed start_steps.  The deques live across episodes (t_queue restarts at 1, so no window spans a reset)."""
    from collections import deque
    import numpy as np
    if rng is None:
        rng = np.random
    if make_agent is None:
        from .agent import Actor
        make_agent = lambda o_: Actor(o_, job="worker")
    if make_env is None:
        from . import env as _env
        make_env = lambda: _env.Wrapper(_default_env(opt.env_name, opt), opt.obs_noise, opt.act_noise, opt.reward_scale, 3)
    agent = make_agent(opt)
    keys = agent.get_weights()[0]
    filling_steps = 0
    env = make_env()
    o_queue = deque([], maxlen=opt.Ln + 1)
    a_r_d_queue = deque([], maxlen=opt.Ln)
    o, r, d, ep_ret, ep_len = env.reset(), 0, False, 0, 0
    t_queue = 1
    o_queue.append((o,))
    weights = _get(_remote(ps.pull, keys))
    agent.set_weights(keys, weights)
    while not _stop(opt):
        if filling_steps > opt.start_steps or opt.weights_file:
            a = agent.get_action(o, deterministic=False)
        else:
            a = env.action_space.sample()
            filling_steps += 1
        o2, r, d, _ = env.step(a)
        ep_ret += r
        ep_len += 1
        o = o2
        a_r_d_queue.append((a, r, d,))
        o_queue.append((o2,))
        if t_queue >= opt.Ln and t_queue % opt.save_freq == 0:
            # Ray pickles the arguments at .remote() time; an in-process actor handle would iterate the live deques later, while this
            # loop keeps appending — so the call takes snapshots
            _remote(replay_buffer[rng.choice(opt.num_buffers, 1)[0]].store, tuple(o_queue), tuple(a_r_d_queue), worker_index)
        t_queue += 1
        if d or (ep_len * opt.action_repeat >= opt.max_ep_len):
            sample_times, steps, _ = _get(_remote(replay_buffer[0].get_counts))
            if steps > opt.start_steps:
                weights = _get(_remote(ps.pull, keys))
                agent.set_weights(keys, weights)
            o, r, d, ep_ret, ep_len = env.reset(), 0, False, 0, 0
            t_queue = 1
            o_queue.append((o,))


def worker_train_nstep(ps, replay_buffer, opt, learner_index, make_agent=None, make_cache=None, push_every=100):
    """algos/sac1/sac_ray.py:155-175: the learner behind the Cache helper, weights into its q2 every 100 updates (the literal at :173)."""
    if make_agent is None:
        from .agent import Learner
        make_agent = lambda o_: Learner(o_, job="learner")
    if make_cache is None:
        make_cache = lambda rb: BatchCache(rb, opt, [ps], nodes=False)
    agent = make_agent(opt)
    keys = agent.get_weights()[0]
    weights = _get(_remote(ps.pull, keys))
    agent.set_weights(keys, weights)
    cache = make_cache(replay_buffer)
    cache.start()
    max_updates = getattr(opt, "max_updates", None)
    cnt = 1
    while True:
        batch = cache.q1.get()
        agent.train(batch, cnt)
        if cnt % push_every == 0:
            cache.q2.put(agent.get_weights())
        if (max_updates is not None and cnt >= max_updates) or _stop(opt):
            end = getattr(cache, "end", None)
            if end is not None:
                end()
            return cnt
        cnt += 1


# ------------------------------------------------------------------------------------------
# device style: everything stays in HBM (plain objects, not actor handles)
# ------------------------------------------------------------------------------------------
class RolloutDevice:
    """State of one vectorised rollout worker: `opt.num_envs` envs, one Actor, a local replay
    shard and a ParameterServer (or a comm.ParamBroadcast on multi-GPU runs).

    Weight adoption (`opt.adopt`, default "episode"): the reference runs one worker per env and each pulls the server's weights at
    ITS OWN episode end (example/dsac.py:127-130), so at any time the envs act on different versions.  "episode" reproduces that
    exactly on the fused path: the actor keeps min(num_envs, max_ep_len) + 2 resident policy versions, pull() stores what the
    server holds as the newest one, and the env-step kernel moves an env to it where that env's episode ends — the vector of envs
    is then indistinguishable from num_envs reference workers stepped in lock step.  "step" (and shapes outside the fused
    envelope) swaps the weights of every env at the next vector step: never staler than the reference, not identical to it."""

    def __init__(self, ps, replay_buffer, opt, worker_index=0):
        import torch
        from .agent import Actor
        from .env import VecLunarLander
        self.ps, self.rb, self.opt = ps, replay_buffer, opt
        self.env = VecLunarLander(opt.num_envs, seed=int(opt.seed) + 1000003 * int(worker_index),
                                  max_ep_len=opt.max_ep_len)
        self.actor = Actor(opt, job="worker", max_rows=opt.num_envs, index=worker_index)
        self.span = ps.span(self.actor.keys) if ps is not None else None
        self._layout = getattr(ps, "layout", 0)
        self.version = -1
        self.t = 0
        self.o = torch.empty_like(self.env.obs)
        self.act = torch.empty(opt.num_envs, 2, dtype=torch.float32, device=self.env.device)
        self._fused = None if getattr(opt, "fused_rollout", True) else False
        self._fused_live = False
        self.adopt = getattr(opt, "adopt", "episode")
        assert self.adopt in ("episode", "step"), self.adopt
        self._versions = False
        self.auto_pull = True     # step() looks at the server itself; FreeRunningLoop turns it off and pulls at segment boundaries
        self.pull()
        if self.adopt == "episode" and self._fused is None:
            self._fused_ready()   # the version store starts from the initial pull (dsac.py:88-90), before any step

    def pull(self):
        """ps.pull(keys) + agent.set_weights when the server has something newer."""
        if self.ps is None or self.ps.version == self.version:
            return False
        self.version = self.ps.version
        layout = getattr(self.ps, "layout", 0)
        if layout != self._layout:          # a key entered or left the server's flat table since the span was looked up
            self.span, self._layout = self.ps.span(self.actor.keys), layout
        if self.span is not None:
            import torch
            flat = getattr(self.ps, "flat", None)
            if torch.is_tensor(flat):   # same device, same stream: the pack kernel reads the server's buffer in place (its copy IS the snapshot)
                self.actor.set_weights_flat(flat[self.span[0]:self.span[0] + self.span[1]])
            else:
                self.actor.set_weights_flat(self.ps.pull_flat(*self.span))
        else:
            self.actor.set_weights(self.actor.keys, self.ps.pull_device(self.actor.keys))
        return True

    def _fused_ready(self):
        """Can this worker take the fused launch pair (ddrl_rollout_step)?  Decided once: the actor's shape must be inside
        the direct-operand envelope and the ring must be the plain five-array SAC layout."""
        if self._fused is None:
            from . import _lib
            from .replay import ReplayBuffer
            ok = isinstance(self.rb, ReplayBuffer) and not getattr(self.rb, "_acts_1d", False) and self.env.n % 32 == 0
            if ok:
                ok = self.actor._lib.ddrl_rollout_begin(self.env._h, self.actor._h, _lib.stream_ptr()) == 0
            self._fused, self._fused_live = ok, ok
            if ok and self.adopt == "episode" and self.actor.max_rows == self.env.n:
                # every env on the weights of the initial pull; later pulls become versions adopted at episode ends
                self.actor.enable_versions(min(2048, min(self.env.n, int(self.opt.max_ep_len)) + 2))
                self._versions = True
        return self._fused

    def step(self, n_steps=1):
        """One vector step = num_envs reference iterations (dsac.py:96-130); n_steps > 1 issues that many back to back with the
        weights the actor holds (the reference pulls at episode ends only).  Policy phase: ONE policy-forward launch + ONE
        launch that finishes get_action, steps the physics and appends the transitions to the ring (ddrl_rollout_step);
        random-action phase (t <= start_steps) and shapes outside the envelope: get_action / env.step / store launches."""
        from . import _lib
        env = self.env
        if self._versions and self.auto_pull:
            self.pull()   # what the server holds NOW is what an env ending its episode in this step pulls (dsac.py:127-130)
        if self.t > self.opt.start_steps and self._fused_ready():
            if not self._fused_live:   # the unfused path has stepped the envs since: refresh the actor's observation rows
                _lib.check(self.actor._lib.ddrl_rollout_begin(env._h, self.actor._h, _lib.stream_ptr()))
                self._fused_live = True
            a = self.actor
            _lib.check(a._lib.ddrl_rollout_step(env._h, a._h, self.rb._h, int(n_steps), a._noise_seed, a._noise_ctr, 0, _lib.dptr(self.act),
                                                _lib.dptr(env.obs), _lib.stream_ptr()))
            a._noise_ctr += int(n_steps) * env.n * a.cfg.act_dim
            self.t += int(n_steps)
            if not self._versions and self.auto_pull:
                self.pull()
            return
        if n_steps > 1:
            for _ in range(int(n_steps)):
                self.step()
            return
        self._fused_live = False
        self.o.copy_(env.obs)
        if self.t > self.opt.start_steps:
            self.actor.get_actions(self.o, out=self.act)
        else:
            env.sample_actions(out=self.act)
        o2, r, d, _, ended = env.step(self.act)
        if self._versions:
            self.actor.adopt_where_ended(ended)
        self.rb.store_batch(self.o, self.act, r, o2, d)
        self.t += 1
        if not self._versions and self.auto_pull:
            self.pull()


class WindowQueue:
    """Per-env o_queue / a_r_d_queue of algos/sac1/sac_ray.py:192-248 on the device (csrc/winq.hip)."""

    def __init__(self, n_envs, Ln, obs_dim, act_dim, save_freq=1, device=None):
        import ctypes
        import torch
        from . import _lib
        _lib.require_gpu()
        self._lib = _lib.load()
        self._check, self._dptr, self._stream = _lib.check, _lib.dptr, _lib.stream_ptr
        self.n, self.Ln, self.obs_dim, self.act_dim = int(n_envs), int(Ln), int(obs_dim), int(act_dim)
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        h = ctypes.c_void_p()
        _lib.check(self._lib.ddrl_winq_create(ctypes.byref(h), self.device.index, self.n, self.Ln, self.obs_dim, self.act_dim, int(save_freq)))
        self._h = h
        p = (ctypes.c_void_p * 4)()
        tq = ctypes.c_void_p()
        _lib.check(self._lib.ddrl_winq_buffers(self._h, p, ctypes.byref(tq)))
        from .replay import _view
        shapes = [(self.n, self.Ln + 1, self.obs_dim), (self.n, self.Ln, self.act_dim), (self.n, self.Ln), (self.n, self.Ln)]
        self.arrays = [_view(p[j], shapes[j], self.device) for j in range(4)]   # o, a, r, d windows: views of the device arrays
        self.ready = torch.zeros(self.n, dtype=torch.uint8, device=self.device)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.ddrl_winq_destroy(h)

    def begin(self, obs, mask=None):
        self._check(self._lib.ddrl_winq_begin(self._h, self._dptr(mask), self._dptr(obs), self._stream()))

    def push(self, obs2, act, rew, done):
        """Returns the device mask (uint8[n]) of the envs whose window is to be stored this step."""
        self._check(self._lib.ddrl_winq_push(self._h, self._dptr(obs2), self._dptr(act), self._dptr(rew), self._dptr(done),
                                             self._dptr(self.ready), self._stream()))
        return self.ready


class RolloutDeviceNStep:
    """worker_rollout of the n-step driver (algos/sac1/sac_ray.py:179-262) for `opt.num_envs` envs per launch:
    Wrapper'd env.step (action noise, action repeat 3, observation noise, reward scale), the per-env window
    queues, and `replay_buffer[random shard].store(o_queue, a_r_d_queue)` for every env whose queues are full —
    one masked row store per vector step.  `replay_buffers` is a ReplayBufferNStep or a list of them."""

    WRAPPER_REPEAT = 3  # the literal in sac_ray.py:189 (opt.action_repeat only scales the episode limit)

    def __init__(self, ps, replay_buffers, opt, worker_index=0):
        import numpy as np
        import torch
        from .agent import Actor
        from .env import VecLunarLander
        self.ps, self.opt = ps, opt
        self.rbs = list(replay_buffers) if isinstance(replay_buffers, (list, tuple)) else [replay_buffers]
        n = int(opt.num_envs)
        self.env = VecLunarLander(n, seed=int(opt.seed) + 1000003 * int(worker_index), max_ep_len=1 << 23)
        self.limit_steps = -(-int(opt.max_ep_len) // int(opt.action_repeat))   # ep_len * action_repeat >= max_ep_len (sac_ray.py:252)
        self.actor = Actor(opt, job="worker", max_rows=n, index=worker_index) if ps is not None else None
        self.span = ps.span(self.actor.keys) if ps is not None else None
        self._layout = getattr(ps, "layout", 0)
        self.version = -1
        self.filling_steps = 0
        self.winq = WindowQueue(n, opt.Ln, opt.obs_dim, opt.act_dim, getattr(opt, "save_freq", 1), device=self.env.device.index)
        self.winq.begin(self.env.obs)
        self.o = torch.empty_like(self.env.obs)
        self.act = torch.empty(n, 2, dtype=torch.float32, device=self.env.device)
        self.pick = np.random.RandomState(int(opt.seed) + 7919 * int(worker_index))
        self.pull()
        # Weight adoption as in RolloutDevice: "episode" (default) = every env acts on what ITS OWN episode end pulled
        # (sac_ray.py:252-262: at `d or ep_len * action_repeat >= max_ep_len`, and only once the buffer's steps exceed start_steps),
        # through the actor's version store; "step" = every env on the newest weights from the next vector step on.
        self.adopt = getattr(opt, "adopt", "episode")
        self._versions = False
        self._learning = bool(getattr(opt, "weights_file", ""))   # steps > start_steps seen (sticky): pulls happen from then on
        if self.adopt == "episode" and self.actor is not None and n % 32 == 0:
            try:
                self.actor.enable_versions(min(2048, min(n, self.limit_steps) + 2))
                self._versions = True
            except ValueError:       # a policy outside the direct-operand envelope: whole-vector swap
                self._versions = False

    pull = RolloutDevice.pull

    def step(self):
        """One vector step = num_envs iterations of sac_ray.py:208-262."""
        env, opt = self.env, self.opt
        if self._versions:
            self.pull()              # what the server holds now is what an env ending its episode in this step would pull
        self.o.copy_(env.obs)
        if self.actor is not None and (self.filling_steps > opt.start_steps or getattr(opt, "weights_file", "")):
            if self._versions:
                self.actor.get_actions_versioned(self.o, self.limit_steps, out=self.act)
            else:
                self.actor.get_actions(self.o, out=self.act)
        else:
            env.sample_actions(out=self.act)
            self.filling_steps += 1
        o2, r, d, next_obs, ended = env.step_wrapped(self.act, getattr(opt, "act_noise", 0.0), getattr(opt, "obs_noise", 0.0),
                                                     getattr(opt, "reward_scale", 1.0), self.WRAPPER_REPEAT, self.limit_steps)
        ready = self.winq.push(o2, self.act, r, d)
        rb = self.rbs[int(self.pick.choice(len(self.rbs), 1)[0])]   # replay_buffer[np.random.choice(opt.num_buffers, 1)[0]]
        rb.store_masked(*self.winq.arrays, ready)
        if self._versions:
            if not self._learning:   # sac_ray.py:255-259: the worker reads the buffer's counters at its episode end, after its store
                self._learning = self.rbs[0].get_counts()[1] > opt.start_steps   # (a device round trip per step, in the filling phase only)
            if self._learning:
                self.actor.adopt_where_ended(ended)
        self.winq.begin(next_obs, ended)
        if not self._versions:
            self.pull()


class TrainDevice:
    """State of one device-resident learner worker: the hot loop of sac1.py:146-151
    (`batch = sample; agent.train(batch); push every push_freq-th update`) with the sample/train
    iterations enqueued by ddrl_loop_run (graph-captured, no host work per update)."""

    def __init__(self, ps, replay_buffer, opt, learner_index=0, updates_per_graph=16, on_push=None):
        import ctypes
        from . import _lib
        from .agent import Learner
        self.ps, self.rb, self.opt = ps, replay_buffer, opt
        self.agent = Learner(opt, job="learner", index=learner_index)
        self.on_push = on_push  # multi-GPU: comm.ParamBroadcast.sync
        if ps is not None:
            span = ps.span(self.agent.keys)
            if span is not None:
                self.agent.set_weights_flat(ps.pull_flat(*span))
            else:
                self.agent.set_weights(self.agent.keys, ps.pull_device(self.agent.keys))
        self.cnt = 1
        self.push_freq = int(getattr(opt, "push_freq", 300))
        self._lib, self._libmod = _lib.load(), _lib
        h = ctypes.c_void_p()
        seed = (int(getattr(opt, "seed", 0)) * 2654435761 + 97 * int(learner_index) + 1) & 0xFFFFFFFF
        self.noise_seed = seed
        _lib.check(self._lib.ddrl_loop_create(ctypes.byref(h), self.agent._h, replay_buffer._h, int(updates_per_graph), seed))
        self._h = h

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.ddrl_loop_destroy(h)

    def push(self):
        flat = self.agent.get_weights_flat()
        if self.ps is not None:
            self.ps.push_flat(flat)
        if self.on_push is not None:
            self.on_push(flat)

    def run(self, n_updates):
        """n_updates iterations of sample -> train, pushing after every push_freq-th update."""
        left = int(n_updates)
        while left > 0:
            to_push = self.push_freq - ((self.cnt - 1) % self.push_freq)  # updates until cnt % push_freq == 0
            k = min(left, to_push)
            self._libmod.check(self._lib.ddrl_loop_run(self._h, k, self._libmod.stream_ptr()))
            self.cnt += k
            left -= k
            if (self.cnt - 1) % self.push_freq == 0:
                self.push()

    def step(self):
        self.run(1)


class TrainDeviceDQN:
    """Device-resident form of algos/dqn/train.py:213-231 + its Cache (177-210): every update draws from a random buffer of a random
    node (`np.random.choice`, the Cache's draw) — `agent.train_from(buffer)`: the indices come from that buffer's own sampler, the
    layer-1 forward reads the rows where they lie (ddrl_dqn_step_ring) — and the weights go to EVERY node's parameter server after each
    push_freq-th update.  No helper thread: with the buffers in the learner's HBM there is no sample latency left to hide.
    `node_buffer[node][buffer]` and `node_ps[node]` are plain (same-process) objects."""

    def __init__(self, node_ps, node_buffer, opt, learner_index=0, make_agent=None, rng=None):
        import numpy as np
        if make_agent is None:
            from .dqn import Learner
            make_agent = lambda o_: Learner(o_, job="learner")
        self.node_ps, self.node_buffer, self.opt = node_ps, node_buffer, opt
        self.agent = make_agent(opt)
        self.keys = self.agent.get_weights()[0]
        self.agent.set_weights(self.keys, node_ps[0].pull(self.keys))
        self.rng = np.random if rng is None else rng
        self.cnt = 1

    def run(self, n_updates):
        opt = self.opt
        for _ in range(int(n_updates)):
            node_idx = self.rng.choice(opt.num_nodes, 1)[0]
            buffer_idx = self.rng.choice(opt.num_buffers, 1)[0]
            self.agent.train_from(self.node_buffer[node_idx][buffer_idx], self.cnt)
            if self.cnt % opt.push_freq == 0:
                keys, values = self.agent.get_weights()
                for ps in self.node_ps:
                    ps.push(keys, values)
            self.cnt += 1
        return self.cnt - 1


class ActorLearnerLoop:
    """The actor/learner ratio gate of algos/sac1/sac1.py:203-207 (`while steps / sample_times > a_l_ratio:
    sleep`) and the `Cache` prefetch of sac1.py:103-130 for the device-resident workers.  The reference
    stalls the rollouts until the learner has caught up; with both on one stream the same invariant —
    steps / sample_times <= a_l_ratio once learning has started (steps > start_steps, sac1.py:196) — is
    kept by running, after every vector step, exactly the updates that re-open the gate.  The prefetch
    queue is the learner's two input sets: update u+1's batch is drawn while update u runs
    (ddrl_sac1_step_and_sample inside ddrl_loop_run).  Counts are tracked on the host (no device sync);
    `counts()` reads the device."""

    def __init__(self, rollout, trainer, opt):
        self.rollout, self.trainer, self.opt = rollout, trainer, opt
        self.a_l_ratio = float(opt.a_l_ratio)
        self.start_steps = int(getattr(opt, "start_steps", 0))
        self.steps = 0          # store() calls (rollout side)
        self.sample_times = 0   # sample_batch() calls (learner side)

    def due(self):
        """Updates the learner owes before the rollouts may continue (0 while the buffer is filling)."""
        if self.steps <= self.start_steps:
            return 0
        return max(0, int(self.steps // self.a_l_ratio) - self.sample_times)

    def run(self, vector_steps):
        n_env = int(self.opt.num_envs)
        for _ in range(int(vector_steps)):
            self.rollout.step()
            self.steps += n_env
            k = self.due()
            if k:
                self.trainer.run(k)
                self.sample_times += k

    def counts(self):
        return self.trainer.rb.get_counts()


_FREE_STREAMS = {}


def _free_streams(torch):
    dev = torch.cuda.current_device()
    if dev not in _FREE_STREAMS:
        _FREE_STREAMS[dev] = (torch.cuda.Stream(), torch.cuda.Stream())
    return _FREE_STREAMS[dev]


class FreeRunningLoop:
    """example/dsac.py:229-236 starts its rollout and learner tasks and leaves them running — `worker_rollout` (dsac.py:76-130) and
    `worker_train` (dsac.py:133-150) have NO gate between them: nothing couples them but the replay buffer's store / sample_batch
    calls and the parameter server's push / pull.  Device form of that asynchrony on ONE GPU: the vectorised rollout free-runs on
    its own HIP stream while the learner's graph loop trains on another, in segments of `steps_per_segment` vector steps beside
    `updates_per_segment` updates.

    What crosses between the two streams crosses at segment boundaries, behind events — kernels of two streams that overlap in
    time have no coherent view of each other's stores on this part (per-XCD L2s), so nothing is shared while it is being written:
      * store: the rollout appends to one of two STAGING rings of exactly steps_per_segment x num_envs rows (a vector step is still
        `num_envs` reference store() calls, in env order); when the learner's segment is over and the rollout's staging ring is
        complete, the learner's stream commits it to the replay ring with one store_batch — the same rows in the same order the
        direct stores would have written, `steps` advancing by the same count.  The reference's store is a fire-and-forget RPC
        (dsac.py:112): a transition reaches the buffer some queueing delay after the env step that produced it; here that delay
        is at most one segment.
      * pull: the rollout adopts what the server held at the END of the learner's previous segment (per env at that env's own
        episode end, through the actor's version store, exactly as in RolloutDevice); the learner's pushes of the running segment
        are not looked at until the boundary.
    Neither side ever waits for the other INSIDE a segment; at a boundary the rollout waits for the learner's previous segment
    (its pushes) and the commit waits for the rollout's segment.  With both segment lengths about equal in time both streams stay
    busy.  The index streams stay exactly the replay ring's own (one MT19937 draw per update, in update order); which transitions
    the ring holds at a given update is timing-free too (commits sit at fixed points of the update sequence), so a run is
    reproducible — unlike the reference's."""

    def __init__(self, rollout, trainer, opt, steps_per_segment=64, updates_per_segment=None, timing=False, streams=None):
        import torch
        from .replay import ReplayBufferSAC1
        self.rollout, self.trainer, self.opt = rollout, trainer, opt
        self.timing, self.marks = bool(timing), []
        self.K = int(steps_per_segment)
        self.n = int(self.K if updates_per_segment is None else updates_per_segment)
        self.rb = trainer.rb
        n_env = int(opt.num_envs)
        self.rows = self.K * n_env
        assert self.rows <= self.rb.max_size, "a segment's transitions must fit the replay ring"
        self.stage = [ReplayBufferSAC1(opt.obs_dim, opt.act_dim, self.rows) for _ in range(2)]
        self.views = [st.rings() for st in self.stage]
        # streams=(s, s): both halves on ONE stream — the same launches in the order [pull, K vector steps, n updates, commit] per
        # segment: what the two-stream run must equal bit for bit (tests)
        # (the pair is created once per device and reused: the runtime multiplexes streams onto a handful of hardware queues —
        # four by default — and a later pair can land on ONE queue, where the two halves then run strictly one after the other;
        # measured: the second FreeRunningLoop of a process took 9.7 ms per segment against the first one's 6.1)
        self.sR, self.sL = streams if streams is not None else _free_streams(torch)
        self.seg = 0
        self.ev_roll, self.ev_commit, self.ev_learn, self.ev_pull = {}, {}, {}, {}
        self.env_steps = self.updates = 0
        rollout.auto_pull = False
        # both streams start behind whatever the caller's stream has queued (construction, ring fill, set_weights)
        ev = torch.cuda.Event()
        ev.record()
        self.sR.wait_event(ev)
        self.sL.wait_event(ev)

    def run(self, segments):
        import torch
        ro, tr = self.rollout, self.trainer
        mk = (lambda: torch.cuda.Event(enable_timing=True)) if self.timing else None
        for _ in range(int(segments)):
            s = self.seg
            m = {}
            with torch.cuda.stream(self.sR):
                if s - 1 in self.ev_learn:
                    self.sR.wait_event(self.ev_learn[s - 1])      # the pushes of the learner's previous segment have landed
                if mk:
                    m["r0"] = mk(); m["r0"].record()
                ro.pull()
                self.ev_pull[s] = torch.cuda.Event()
                self.ev_pull[s].record()
                if s - 2 in self.ev_commit:
                    self.sR.wait_event(self.ev_commit.pop(s - 2))  # this staging ring's last commit has read it
                ro.rb = self.stage[s % 2]
                ro.step(self.K)
                self.ev_roll[s] = mk() if mk else torch.cuda.Event()
                self.ev_roll[s].record()
                m["r1"] = self.ev_roll[s]
            with torch.cuda.stream(self.sL):
                self.sL.wait_event(self.ev_pull.pop(s))            # (the pack kernel of that pull reads the server's buffer)
                if mk:
                    m["l0"] = mk(); m["l0"].record()
                tr.run(self.n)
                self.ev_learn[s] = mk() if mk else torch.cuda.Event()
                self.ev_learn[s].record()
                m["l1"] = self.ev_learn[s]
                self.sL.wait_event(self.ev_roll.pop(s))
                if mk:
                    m["c0"] = mk(); m["c0"].record()
                v = self.views[s % 2]
                self.rb.store_batch(v["obs1_buf"], v["acts_buf"], v["rews_buf"], v["obs2_buf"], v["done_buf"])
                self.ev_commit[s] = mk() if mk else torch.cuda.Event()
                self.ev_commit[s].record()
                m["c1"] = self.ev_commit[s]
            if mk:
                self.marks.append(m)
            self.ev_learn.pop(s - 1, None)
            self.seg += 1
            self.env_steps += self.rows
            self.updates += self.n

    def drain(self):
        """Both streams empty (and the caller's stream ordered behind them)."""
        import torch
        self.sR.synchronize()
        self.sL.synchronize()
        torch.cuda.current_stream().synchronize()

    def segment_times(self, last=8):
        """timing=True: mean device time (ms) of the last segments' phases — the rollout's K vector steps, the learner's n updates, the
        commit — and the segment's span on the learner stream (start of its updates to the end of its commit)."""
        self.drain()
        ms = self.marks[-int(last):]
        if not ms:
            return None
        avg = lambda a, b: sum(m[a].elapsed_time(m[b]) for m in ms) / len(ms)
        return {"rollout_ms": avg("r0", "r1"), "learner_ms": avg("l0", "l1"), "commit_ms": avg("c0", "c1"), "segment_ms": avg("l0", "c1"),
                "segments": len(ms)}

    def close(self):
        """Back to one stream: the rollout stores into the replay ring again and looks at the server itself."""
        self.drain()
        self.rollout.rb, self.rollout.auto_pull = self.rb, True
