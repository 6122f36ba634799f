"""Host logic (no GPU): the DQN / SQN driver's workers (algos/dqn/train.py:177-371) against traces recorded from the reference's OWN
functions on scripted fakes (tests/golden/dqn_driver_traces.json, oracle/gen_golden.py::gen_dqn_driver_traces): event order of
worker_rollout (random buffer per store, counters of one random buffer per episode, policy / random switch, recover),
worker_train (Cache hand-off, push cadence), the Cache helper's loop, worker_test (counters, a_l_ratio, update frequency, TensorBoard
values, weight pickles, checkpoint fan-out)."""
import json
import os

import numpy as np
import pytest

from distributed_drl_amd import workers


class _Stop(Exception):
    pass


class _Args:
    pass


@pytest.fixture(scope="module")
def traces(golden_dir):
    return json.load(open(os.path.join(golden_dir, "dqn_driver_traces.json")))


class _Rng:
    """np.random as the reference's worker sees it: seed() and choice(n, 1), scripted."""

    def __init__(self, ev, choices):
        self.ev, self.choices = ev, list(choices)

    def seed(self, *a):
        self.ev.append(["np_seed"])

    def choice(self, n, k):
        v = self.choices.pop(0) if self.choices else 0
        self.ev.append(["choice", int(n), int(v)])
        return np.array([v])


def _fakes(ev, lens):
    class FakeSpace:
        def sample(self):
            ev.append(["sample_random"])
            return 2

    class FakeEnv:
        rewards = [7.5]

        def __init__(self):
            self.action_space = FakeSpace()
            self.ep, self.k = -1, 0

        def reset(self):
            self.ep += 1
            if self.ep >= len(lens):
                raise _Stop()
            self.k = 0
            ev.append(["reset"])
            return np.full(4, float(self.ep), np.float64)

        def step(self, a):
            self.k += 1
            ev.append(["env_step", int(a)])
            return np.full(4, self.ep + 0.01 * self.k, np.float64), 0.5 * self.k, self.k >= lens[self.ep], {}

    class FakeAgent:
        def __init__(self, job):
            ev.append(["agent", job])

        def get_weights(self):
            return ["main/q/w"], [np.zeros(1, np.float32)]

        def set_weights(self, keys, w):
            ev.append(["set_weights", list(keys)])

        def get_action(self, o):
            ev.append(["get_action", float(np.asarray(o).ravel()[0])])
            return 1

        def train(self, batch, cnt):
            ev.append(["train", int(batch["id"]), int(cnt)])
            if cnt >= 7:
                raise _Stop()

    class FakePS:
        def __init__(self, idx=0):
            self.idx = idx

        def pull(self, keys):
            ev.append(["pull", list(keys)])
            return [np.zeros(1, np.float32)]

        def push(self, keys, vals):
            ev.append(["push", self.idx, list(keys)])

        def get_weights(self):
            ev.append(["ps_get_weights"])
            return {"main/q/w": np.zeros(1, np.float32), "target/q/w": np.ones(1, np.float32)}

        def save_weights(self):
            ev.append(["ps_save_weights", self.idx])
            return "psop"

    class FakeRB:
        def __init__(self, name, counts):
            self.name, self.counts = name, [list(c) for c in counts]

        def store(self, o, a, r, o2, d, wi):
            ev.append(["store", self.name, float(np.asarray(o).ravel()[0]), int(a), float(r), float(np.asarray(o2).ravel()[0]), bool(d), int(wi)])

        def get_counts(self):
            c = self.counts.pop(0) if len(self.counts) > 1 else self.counts[0]
            ev.append(["get_counts", self.name, list(c)])
            return tuple(c)

        def sample_batch(self):
            ev.append(["sample_batch", self.name])
            return {"id": 0}

        def save(self):
            ev.append(["rb_save", self.name])
            return "rbop"

    return FakeEnv, FakeAgent, FakePS, FakeRB


def test_worker_rollout_dqn_event_order_matches_reference(traces):
    t = traces["rollout"]
    ev = []
    Env, Agent, PS, RB = _fakes(ev, t["args"]["episode_lens"])
    opt = _Args()
    opt.num_buffers, opt.start_steps, opt.recover = t["args"]["num_buffers"], t["args"]["start_steps"], t["args"]["recover"]
    rbs = [RB(k, t["args"]["counts"][k]) for k in ("b0", "b1")]
    with pytest.raises(_Stop):
        workers.worker_rollout_dqn(PS(), rbs, opt, t["args"]["worker_index"], make_env=Env, make_agent=lambda o: Agent("worker"),
                                   rng=_Rng(ev, t["args"]["choices"]))
    assert ev == t["events"]
    # the switch reads ONE random buffer's actor_steps, scaled by num_buffers, once per episode: episode 2 (4 * 2 = 8 <= 10) is still random
    assert sum(1 for e in ev if e[0] == "sample_random") == 5 and sum(1 for e in ev if e[0] == "get_action") == 4
    # ... and with opt.recover the policy acts from the first step whatever the counters say
    ev.clear()
    Env, Agent, PS, RB = _fakes(ev, [2])
    opt.recover = True
    with pytest.raises(_Stop):
        workers.worker_rollout_dqn(PS(), [RB("b0", [[0, 0, 0]])], opt, 0, make_env=Env, make_agent=lambda o: Agent("worker"), rng=_Rng(ev, [0, 0, 0, 0]))
    assert ev == traces["rollout_recover"]["events"]


def test_worker_train_dqn_hands_weights_to_the_cache_every_push_freq(traces):
    t = traces["train"]
    ev = []
    Env, Agent, PS, RB = _fakes(ev, [1])

    class Q:
        def __init__(self, name):
            self.name, self.n = name, 0

        def get(self):
            self.n += 1
            return {"id": self.n}

        def put(self, x):
            ev.append(["%s_put" % self.name, list(x[0])])

    class Cache:
        def __init__(self, nb):
            ev.append(["cache_init"])
            self.q1, self.q2 = Q("q1"), Q("q2")

        def start(self):
            ev.append(["cache_start"])

    opt = _Args()
    opt.push_freq = t["args"]["push_freq"]
    with pytest.raises(_Stop):
        workers.worker_train_dqn(PS(), [[None]], opt, 0, make_agent=lambda o: Agent("learner"), make_cache=Cache)
    assert ev == t["events"]


def test_batch_cache_loop_matches_the_reference_helper(traces):
    t = traces["cache"]
    ev = []
    Env, Agent, PS, RB = _fakes(ev, [1])
    opt = _Args()
    opt.num_nodes, opt.num_buffers = t["args"]["num_nodes"], t["args"]["num_buffers"]
    sizes, empties = list(t["args"]["q1_sizes"]), list(t["args"]["q2_empty"])

    class Q1:
        def qsize(self):
            if not sizes:
                raise _Stop()
            v = sizes.pop(0)
            ev.append(["q1_qsize", v])
            return v

        def put(self, b):
            ev.append(["q1_put"])

    class Q2:
        def empty(self):
            v = empties.pop(0)
            ev.append(["q2_empty", v])
            return v

        def get(self):
            ev.append(["q2_get"])
            return ["main/q/w"], [np.zeros(1, np.float32)]

    nb = [[RB("n%db%d" % (n, b), [[0, 0, 0]]) for b in range(2)] for n in range(2)]
    cache = workers.BatchCache(nb, opt, [PS(0), PS(1)], rng=_Rng(ev, t["args"]["choices"]))
    with pytest.raises(_Stop):
        cache.ps_update(Q1(), Q2(), nb)
    assert ev == t["events"]


def test_batch_cache_thread_feeds_a_real_learner_loop():
    """The thread form end to end on plain objects: batches arrive, pushes reach every node's server, end() stops the helper."""
    ev = []
    Env, Agent, PS, RB = _fakes(ev, [1])
    opt = _Args()
    opt.num_nodes, opt.num_buffers, opt.push_freq, opt.max_updates = 2, 2, 2, 6
    nb = [[RB("n%db%d" % (n, b), [[0, 0, 0]]) for b in range(2)] for n in range(2)]
    pss = [PS(0), PS(1)]

    class Learner(Agent):
        def train(self, batch, cnt):
            pass
    n = workers.worker_train_dqn(pss[0], nb, opt, 0, node_ps=pss, make_agent=lambda o: Learner("learner"))
    assert n == 6
    pushes = [e for e in ev if e[0] == "push"]
    assert len(pushes) == 6 and [e[1] for e in pushes] == [0, 1] * 3   # three hand-offs, each to both servers; end() lets the last ones out
    assert sum(1 for e in ev if e[0] == "sample_batch") >= 6


def test_worker_test_dqn_matches_reference(traces, tmp_path):
    t = traces["test"]
    ev = []
    Env, Agent, PS, RB = _fakes(ev, [1])
    results = [tuple(r) for r in t["args"]["results"]]

    class Tester(Agent):
        def test(self, env, n):
            if not results:
                raise _Stop()
            ev.append(["agent_test", int(n)])
            return results.pop(0)

        def write_tb(self, ave_test_reward, ave_score, alratio, update_frequency, total_learner_step):
            ev.append(["write_tb", float(ave_test_reward), float(ave_score), float(alratio), int(update_frequency), int(total_learner_step)])

    clock = list(t["args"]["clock"])

    def now():
        v = clock.pop(0) if len(clock) > 1 else clock[0]
        ev.append(["time", v])
        return v
    opt = _Args()
    opt.num_nodes, opt.num_buffers, opt.save_interval = t["args"]["num_nodes"], t["args"]["num_buffers"], t["args"]["save_interval"]
    opt.checkpoint_freq, opt.save_dir = t["args"]["checkpoint_freq"], str(tmp_path)
    nb = [[RB("b0", [[20, 100, 100], [600, 300, 300], [600, 300, 300], [1300, 500, 500]]),
           RB("b1", [[20, 100, 100], [650, 320, 320], [650, 320, 320], [1300, 520, 520]])]]
    ps = PS(0)
    with pytest.raises(_Stop):
        workers.worker_test_dqn(ps, nb, opt, node_ps=[ps], make_env=Env, make_agent=lambda o: Tester("test"), clock=now, log=lambda s: None,
                                wait=lambda ops, num_returns: ev.append(["ray_wait", len(ops), int(num_returns)]))
    assert ev == t["events"]
    assert sorted(os.listdir(tmp_path)) == t["files"]
    import pickle
    w = pickle.load(open(tmp_path / t["files"][0], "rb"))
    assert sorted(w) == ["main/q/w", "target/q/w"]                    # the WHOLE weight dict, not only the actor's keys


def test_get_al_status_orders_like_the_reference():
    ev = []
    Env, Agent, PS, RB = _fakes(ev, [1])
    opt = _Args()
    opt.num_nodes, opt.num_buffers = 2, 2
    nb = [[RB("n0b0", [[1, 2, 3]]), RB("n0b1", [[4, 5, 6]])], [RB("n1b0", [[7, 8, 9]]), RB("n1b1", [[10, 11, 12]])]]
    actor, learner, size = workers.get_al_status(nb, opt)
    assert actor.tolist() == [2, 5, 8, 11] and learner.tolist() == [1, 4, 7, 10] and size.tolist() == [3, 6, 9, 12]


# ---- the n-step driver (algos/sac1/sac_ray.py:123-274) -------------------------------------------------------------------------
@pytest.fixture(scope="module")
def ntraces(golden_dir):
    return json.load(open(os.path.join(golden_dir, "nstep_driver_traces.json")))


def _nstep_fakes(ev, lens):
    class FakeSpace:
        def sample(self):
            ev.append(["sample_random"])
            return np.array([0.5, -0.5], np.float32)

    class FakeEnv:
        def __init__(self):
            self.action_space = FakeSpace()
            self.ep, self.k = -1, 0

        def reset(self):
            self.ep += 1
            if self.ep >= len(lens):
                raise _Stop()
            self.k = 0
            ev.append(["reset"])
            return np.full(3, float(self.ep), np.float64)

        def step(self, a):
            self.k += 1
            ev.append(["env_step", [float(x) for x in np.asarray(a).ravel()]])
            return np.full(3, self.ep + 0.01 * self.k, np.float64), 0.25 * self.k, self.k >= lens[self.ep], {}

    class FakeAgent:
        def __init__(self, job):
            ev.append(["agent", job])

        def get_weights(self):
            return ["main/pi/w"], [np.zeros(1, np.float32)]

        def set_weights(self, keys, w):
            ev.append(["set_weights", list(keys)])

        def get_action(self, o, deterministic=False):
            ev.append(["get_action", float(np.asarray(o).ravel()[0]), bool(deterministic)])
            return np.array([0.1, 0.2], np.float32)

        def train(self, batch, cnt):
            if cnt % 100 == 0 or cnt == 1:
                ev.append(["train", int(batch["id"]), int(cnt)])
            if cnt >= 201:
                raise _Stop()

    class FakePS:
        def pull(self, keys):
            ev.append(["pull", list(keys)])
            return [np.zeros(1, np.float32)]

        def push(self, keys, vals):
            ev.append(["push", list(keys)])

    class FakeRB:
        def __init__(self, name, counts):
            self.name, self.counts = name, [list(c) for c in counts]

        def store(self, oq, aq, wi):
            ev.append(["store", self.name, [float(np.asarray(q[0]).ravel()[0]) for q in oq], [[float(q[0][0]), float(q[1]), bool(q[2])] for q in aq], int(wi)])

        def get_counts(self):
            c = self.counts.pop(0) if len(self.counts) > 1 else self.counts[0]
            ev.append(["get_counts", self.name, list(c)])
            return tuple(c)

        def sample_batch(self):
            ev.append(["sample_batch", self.name])
            return {"id": 0}

    return FakeEnv, FakeAgent, FakePS, FakeRB


def test_worker_rollout_nstep_event_order_matches_reference(ntraces):
    t = ntraces["rollout"]
    a = t["args"]
    ev = []
    Env, Agent, PS, RB = _nstep_fakes(ev, a["episode_lens"])
    opt = _Args()
    opt.Ln, opt.save_freq, opt.num_buffers, opt.start_steps, opt.weights_file = a["Ln"], a["save_freq"], a["num_buffers"], a["start_steps"], ""
    opt.action_repeat, opt.max_ep_len = a["action_repeat"], a["max_ep_len"]
    with pytest.raises(_Stop):
        workers.worker_rollout_nstep(PS(), [RB("b0", a["counts_b0"]), RB("b1", [[0, 0, 0]])], opt, a["worker_index"], make_env=Env,
                                     make_agent=lambda o: Agent("worker"), rng=_Rng(ev, a["choices"]))
    assert ev == t["events"]
    # start_steps + 1 random actions (the budget counts random actions only); the first episode's end does not pull (buffer 0 at 3 <= 4 steps)
    assert sum(1 for e in ev if e[0] == "sample_random") == a["start_steps"] + 1
    first_counts = ev.index(["get_counts", "b0", [0, 3, 3]])
    assert ev[first_counts + 1] == ["reset"]


def test_worker_train_nstep_pushes_every_hundred(ntraces):
    ev = []
    Env, Agent, PS, RB = _nstep_fakes(ev, [1])

    class Q:
        def __init__(self, name):
            self.name, self.n = name, 0

        def get(self):
            self.n += 1
            return {"id": self.n}

        def put(self, x):
            ev.append(["%s_put" % self.name, list(x[0])])

    class Cache:
        def __init__(self, rb):
            ev.append(["cache_init"])
            self.q1, self.q2 = Q("q1"), Q("q2")

        def start(self):
            ev.append(["cache_start"])

    with pytest.raises(_Stop):
        workers.worker_train_nstep(PS(), [None], _Args(), 0, make_agent=lambda o: Agent("learner"), make_cache=Cache)
    assert ev == ntraces["train"]["events"]


def test_batch_cache_single_server_form_matches_reference(ntraces):
    t = ntraces["cache"]
    ev = []
    Env, Agent, PS, RB = _nstep_fakes(ev, [1])
    opt = _Args()
    opt.num_buffers = t["args"]["num_buffers"]
    sizes, empties = list(t["args"]["q1_sizes"]), list(t["args"]["q2_empty"])

    class Q1:
        def qsize(self):
            if not sizes:
                raise _Stop()
            v = sizes.pop(0)
            ev.append(["q1_qsize", v])
            return v

        def put(self, b):
            ev.append(["q1_put"])

    class Q2:
        def empty(self):
            v = empties.pop(0)
            ev.append(["q2_empty", v])
            return v

        def get(self):
            ev.append(["q2_get"])
            return ["main/pi/w"], [np.zeros(1, np.float32)]

    rbs = [RB("b0", [[0, 0, 0]]), RB("b1", [[0, 0, 0]])]
    cache = workers.BatchCache(rbs, opt, [PS()], rng=_Rng(ev, t["args"]["choices"]), nodes=False)
    assert cache.q1.maxsize == 10 and cache.q2.maxsize == 5
    with pytest.raises(_Stop):
        cache.ps_update(Q1(), Q2(), rbs)
    assert ev == t["events"]


def test_host_wrapper_branches():
    """algos/sac1/hyperparams.py:107-134 as written: in-place action noise, summed and scaled reward over the repeat, reward 0.0 at a
    terminal inside the repeat, raw first-step reward and un-noised observation for action_repeat == 1."""
    from distributed_drl_amd.env import Wrapper

    class E:
        def __init__(self, end_at):
            self.k, self.end_at = 0, end_at

        def reset(self):
            self.k = 0
            return np.zeros(3)

        def step(self, a):
            self.k += 1
            return np.full(3, float(self.k)), 2.0, self.k >= self.end_at, {}

    rs = np.random.RandomState(0)
    w = Wrapper(E(100), 0.0, 0.5, 5.0, 3, rng=rs)
    a = np.zeros(2)
    o, r, d, _ = w.step(a)
    assert np.abs(a).max() > 0 and np.abs(a).max() <= 0.5 and r == 5.0 * 6.0 and not d and o[0] == 3.0   # the caller's array carries the noise
    w = Wrapper(E(2), 0.0, 0.0, 5.0, 3, rng=rs)
    o, r, d, _ = w.step(np.zeros(2))
    assert d and r == 0.0 and o[0] == 2.0
    w = Wrapper(E(100), 0.3, 0.0, 5.0, 1, rng=rs)
    o, r, d, _ = w.step(np.zeros(2))
    assert r == 2.0 and (o == 1.0).all()                                 # neither scaled nor noised
    assert np.abs(w.reset()).max() <= 0.3 and np.abs(w.reset()).max() > 0


def test_dqn_hyperparameters_mirror_the_reference_defaults():
    """algos/dqn/hyperparams.py:10-75: the derived entries (buffers per 25 workers, buffer_size and start_steps split over them,
    start_steps = buffer_size when weights are loaded)."""
    from distributed_drl_amd.dqn import HyperParameters

    class Env:
        class observation_space:
            shape = (38,)

        class action_space:
            n = 3
    o = HyperParameters(Env, num_workers=60)
    assert (o.obs_dim, o.act_dim, o.num_buffers, o.buffer_size, o.start_steps) == (38, 3, 3, 333333, 3333)
    assert (o.push_freq, o.gamma, o.lr, o.polyak, o.batch_size, o.hidden_size, o.checkpoint_freq, o.save_interval) == (100, 0.99, 1e-3, 0.995, 128, [400, 300], 21600, 500000)
    assert HyperParameters(Env, weights_file="w.pickle").start_steps == 1000000 and HyperParameters(obs_dim=84 * 84 * 4, act_dim=4).obs_dim == 28224


def test_nstep_rollout_through_actor_handles_stores_the_windows_of_the_call_moment():
    """Ray pickles the arguments when .remote() is called; the in-process actor handle runs the call later, on its
    mailbox thread, while the worker keeps appending to its deques.  Behind a SLOW store the stored windows must still be the ones a
    synchronous run stores (plain buffer objects: the form test_worker_rollout_nstep_event_order_matches_reference pins against the
    reference's own function), in the same order."""
    import time as _time
    from distributed_drl_amd import remote
    lens, choices = [12, 9, 15], [i % 2 for i in range(64)]

    def run(make_buffers, drain):
        got = []

        class Buffer:
            def __init__(self, name, delay):
                self.name, self.delay = name, delay

            def store(self, oq, aq, wi):
                _time.sleep(self.delay)      # the worker runs many env steps ahead of a slow mailbox
                got.append(["store", self.name, [float(np.asarray(q[0]).ravel()[0]) for q in oq],
                            [[float(q[0][0]), float(q[1]), bool(q[2])] for q in aq], int(wi)])

            def get_counts(self):
                return (0, 100, 100)

        ev = []
        Env, Agent, PS, RB = _nstep_fakes(ev, lens)
        opt = _Args()
        opt.Ln, opt.save_freq, opt.num_buffers, opt.start_steps, opt.weights_file = 3, 1, 2, 4, ""
        opt.action_repeat, opt.max_ep_len = 1, 1000
        bufs = make_buffers(Buffer)
        with pytest.raises(_Stop):
            workers.worker_rollout_nstep(PS(), bufs, opt, 5, make_env=Env, make_agent=lambda o: Agent("worker"), rng=_Rng(ev, list(choices)))
        drain(bufs)
        return got

    want = run(lambda B: [B("b0", 0.0), B("b1", 0.0)], lambda bufs: None)
    got = run(lambda B: [remote.remote(B).remote("b0", 0.01), remote.remote(B).remote("b1", 0.01)],
              lambda bufs: [remote.get(h.get_counts.remote()) for h in bufs])
    assert len(want) == sum(n - 2 for n in lens) and want[0][2] == [0.0, 0.01, 0.02, 0.03]
    for name in ("b0", "b1"):              # per buffer in the order of the calls, with the windows of the call moment
        assert [e for e in got if e[1] == name] == [e for e in want if e[1] == name]


def test_batch_cache_helper_sleeps_when_idle():
    """with ten batches waiting and no weights to forward the helper thread must not spin on the GIL."""
    import time as _time
    ev = []
    Env, Agent, PS, RB = _fakes(ev, [1])
    opt = _Args()
    opt.num_nodes, opt.num_buffers = 1, 1
    calls = [0]

    class CountingRB:
        def sample_batch(self):
            calls[0] += 1
            return {"id": calls[0]}

    cache = workers.BatchCache([[CountingRB()]], opt, [PS(0)])
    polls = [0]
    real_qsize = cache.q1.qsize

    def qsize():
        polls[0] += 1
        return real_qsize()
    cache.q1.qsize = qsize
    cache.start()
    _time.sleep(0.5)
    assert calls[0] == 10 and polls[0] < 60          # a spinning helper polls millions of times in half a second
    t0 = _time.perf_counter()
    assert cache.q1.get()["id"] == 1                 # taking a batch wakes it at once
    for _ in range(1000):
        if calls[0] == 11:
            break
        _time.sleep(0.001)
    assert calls[0] == 11 and _time.perf_counter() - t0 < 0.5    # (an Event wake-up: milliseconds; the bound only has to beat a missed wake-up)
    cache.end()
