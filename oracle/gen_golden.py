#!/usr/bin/env python3
"""Generate golden fixtures for the replay / parameter-server / worker hot path.

TEST INFRASTRUCTURE ONLY.  Runs ONLY in the build container, where the reference
checkout is mounted read-only at /root/reference.  It imports the reference's own
Python modules (example/dsac.py, algos/sac1/sac1.py, algos/dqn/train.py) with stub
modules standing in for the third-party packages that are absent from this image
(ray, tensorflow, gym, spinup), drives the *reference's own* classes/functions
(ReplayBuffer, ParameterServer, worker_rollout, worker_train) and records their
inputs/outputs as small data fixtures under tests/golden/.

No reference source (or bytecode) is copied; only input/output vectors are written.
Nothing at test/bench run time reads /root/reference.

The learner MATH fixtures (tests/golden/{sac1,sacv,dqn,sqn}_math.*) come from oracle/gen_golden_math.py, which executes the
reference's learner classes on oracle/tf_shim.py; this script runs it at the end (a separate process: it installs its own
`tensorflow` stand-in).

Usage:  python oracle/gen_golden.py            # rewrites tests/golden/*.npz|*.json
"""
import importlib.util
import json
import os
import sys
import types
import zlib
from unittest import mock

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


# --------------------------------------------------------------------------------------
# stub third-party modules so that the reference's modules import
# --------------------------------------------------------------------------------------
def _install_stubs():
    ray = types.ModuleType("ray")

    def remote(*args, **kwargs):
        # handles both @ray.remote and @ray.remote(num_gpus=1, max_calls=1)
        if len(args) == 1 and not kwargs and (callable(args[0]) or isinstance(args[0], type)):
            return args[0]
        return lambda obj: obj

    ray.remote = remote
    ray.get = lambda x: x
    ray.wait = lambda x, **k: (x, [])
    ray.init = lambda *a, **k: None
    ray.experimental = types.ModuleType("ray.experimental")
    ray.experimental.tf_utils = mock.MagicMock()
    sys.modules["ray"] = ray
    sys.modules["ray.experimental"] = ray.experimental
    sys.modules["ray.experimental.tf_utils"] = ray.experimental.tf_utils
    for name in ("ray.rllib", "ray.rllib.utils", "ray.rllib.utils.compression"):
        sys.modules[name] = mock.MagicMock()
    for name in ("tensorflow", "gym", "gym.spaces", "spinup", "spinup.algos", "spinup.algos.sac",
                 "spinup.algos.sac.core", "spinup.utils", "spinup.utils.logx",
                 "spinup.utils.run_utils", "pandas"):
        sys.modules[name] = mock.MagicMock()


def _load(path, name, extra_path):
    if extra_path not in sys.path:
        sys.path.insert(0, extra_path)
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _mt_fingerprint():
    st = np.random.get_state()
    key = np.asarray(st[1], dtype=np.uint32)
    return {"pos": int(st[2]), "key_crc32": int(zlib.crc32(key.tobytes())), "key_head": [int(v) for v in key[:4]]}


def _transitions(n, obs_dim, act_dim, act_1d=False):
    """Deterministic transitions in the dtypes a gym loop hands to store():
    float64 observations, float32 actions, python-float reward, python bool done."""
    out = []
    for i in range(n):
        obs = (np.arange(obs_dim, dtype=np.float64) * 0.25 + i * 1.000000123 + 1.0 / 3.0)
        obs2 = obs + 0.1 + 1e-9 * i
        if act_1d:
            act = float(i % 5)
        else:
            act = (np.arange(act_dim, dtype=np.float32) - 0.5) * np.float32(0.3) + np.float32(i) * np.float32(0.01)
        rew = float(i) * 0.7 - 3.3333333333333335
        done = bool(i % 4 == 3)
        out.append((obs, act, rew, obs2, done))
    return out


# --------------------------------------------------------------------------------------
def gen_index_streams(dsac):
    """(1) index streams of ReplayBuffer.sample_batch  (example/dsac.py:39-45)."""
    seeds = [0, 1, 12345, 2 ** 32 - 1]
    sizes = [1, 2, 3, 100, 256, 257, 1000, 65536, 10 ** 6, 4 * 10 ** 6]
    cap = max(sizes)
    buf = dsac.ReplayBuffer(1, 1, cap)
    buf.rews_buf[:] = np.arange(cap, dtype=np.float32)  # row value == row index (exact < 2**24)
    rec = {}
    meta = []
    for s in seeds:
        for n in sizes:
            buf.size = n
            np.random.seed(s)
            a = buf.sample_batch(256)["rews"].astype(np.int64)
            fp1 = _mt_fingerprint()
            b = buf.sample_batch(100)["rews"].astype(np.int64)
            fp2 = _mt_fingerprint()
            tag = "s%d_n%d" % (s, n)
            rec[tag + "_a"] = a.astype(np.int32)
            rec[tag + "_b"] = b.astype(np.int32)
            meta.append({"seed": s, "size": n, "after_256": fp1, "after_100": fp2})
    np.savez_compressed(os.path.join(OUT, "index_streams.npz"), **rec)
    with open(os.path.join(OUT, "index_streams.json"), "w") as f:
        json.dump(meta, f, indent=0)


def gen_ring_and_gather(dsac, sac1, dqn):
    """(2)+(3) ring states after n stores and gathers out of them; (8) counts order."""
    rec = {}
    meta = []
    obs_dim, act_dim = 8, 2
    for cap in (7, 256):
        for n in (cap - 1, cap, cap + 1, 3 * cap + 2):
            for flavour, cls in (("dsac", dsac.ReplayBuffer), ("sac1", sac1.ReplayBuffer)):
                buf = cls(obs_dim, act_dim, cap)
                for tr in _transitions(n, obs_dim, act_dim):
                    buf.store(*tr)
                tag = "%s_c%d_n%d" % (flavour, cap, n)
                for k in ("obs1_buf", "obs2_buf", "acts_buf", "rews_buf", "done_buf"):
                    rec[tag + "_" + k] = getattr(buf, k).copy()
                entry = {"flavour": flavour, "cap": cap, "n": n, "ptr": int(buf.ptr), "size": int(buf.size),
                         "counts_before": _jsonable(buf.get_counts()), "batches": []}
                np.random.seed(7)
                for B in (1, 32, 256):
                    d = buf.sample_batch(B)
                    for k, v in d.items():
                        assert v.dtype == np.float32
                        rec["%s_B%d_%s" % (tag, B, k)] = v
                    entry["batches"].append({"B": B, "mt": _mt_fingerprint()})
                entry["counts_after"] = _jsonable(buf.get_counts())
                meta.append(entry)

    # dqn-shaped buffer (algos/dqn/train.py:43-76): acts 1-D, extra worker_index arg, counts order
    class Opt:
        pass
    opt = Opt()
    opt.buffer_size, opt.obs_dim, opt.batch_size = 9, 12, 5
    buf = dqn.ReplayBuffer(opt, 0)
    for tr in _transitions(13, 12, 1, act_1d=True):
        buf.store(*tr, 3)
    tag = "dqn_c9_n13"
    for k in ("obs1_buf", "obs2_buf", "acts_buf", "rews_buf", "done_buf"):
        rec[tag + "_" + k] = getattr(buf, k).copy()
    np.random.seed(11)
    d = buf.sample_batch()
    for k, v in d.items():
        rec[tag + "_B5_" + k] = v
    meta.append({"flavour": "dqn", "cap": 9, "n": 13, "ptr": int(buf.ptr), "size": int(buf.size),
                 "obs_dim": 12, "B": 5, "seed": 11, "counts_after": _jsonable(buf.get_counts()),
                 "mt": _mt_fingerprint()})

    # (4) empty buffer
    buf = dsac.ReplayBuffer(obs_dim, act_dim, 5)
    try:
        buf.sample_batch(4)
        err = None
    except Exception as e:  # noqa
        err = {"type": type(e).__name__, "msg": str(e)}
    meta.append({"flavour": "empty", "error": err})

    np.savez_compressed(os.path.join(OUT, "ring_gather.npz"), **rec)
    with open(os.path.join(OUT, "ring_gather.json"), "w") as f:
        json.dump(meta, f, indent=0)


def _jsonable(x):
    if isinstance(x, tuple):
        return [int(v) for v in x]
    return int(x)


def gen_ps(dsac):
    """(5) ParameterServer snapshot semantics (example/dsac.py:51-73)."""
    keys = ["main/pi/dense/kernel", "main/pi/dense/bias", "main/q1/dense/kernel"]
    vals = [np.arange(6, dtype=np.float32).reshape(2, 3), np.ones(3, np.float32), np.full((2, 2), 7, np.float32)]
    ps = dsac.ParameterServer(keys, vals)
    log = []
    vals[0][0, 0] = 99.0  # mutate the source after construction: PS must not see it
    log.append({"op": "pull_after_src_mutation", "keys": keys[:1], "out": [v.tolist() for v in ps.pull(keys[:1])]})
    new = [np.full(3, 5, np.float32)]
    ps.push(keys[1:2], new)
    new[0][1] = -1.0  # mutate after push
    log.append({"op": "pull_after_push", "keys": [keys[1], keys[0]],
                "out": [v.tolist() for v in ps.pull([keys[1], keys[0]])]})
    ps.push(["extra/key"], [np.zeros(1, np.float32)])  # push of an unknown key adds it
    log.append({"op": "get_weights_keys", "out": list(ps.get_weights().keys())})
    log.append({"op": "pull_order", "keys": [keys[2], "extra/key", keys[1]],
                "out": [v.tolist() for v in ps.pull([keys[2], "extra/key", keys[1]])]})
    with open(os.path.join(OUT, "ps_trace.json"), "w") as f:
        json.dump(log, f, indent=0)


class _Stop(Exception):
    pass


def gen_worker_traces(dsac, sac1):
    """(6) worker_rollout event order, (7) worker_train push cadence, (8) sac1 throttle."""
    ev = []

    class FakeSpace:
        def sample(self):
            ev.append(["sample_random"])
            return np.array([0.5, -0.5], np.float32)

    class FakeEnv:
        # scripted episode lengths: terminal after 3 steps, then time-limit (max_ep_len=5), then 2
        lens = [3, 99, 2, 99]

        def __init__(self):
            self.action_space = FakeSpace()
            self.ep = -1
            self.k = 0

        def reset(self):
            self.ep += 1
            self.k = 0
            ev.append(["reset"])
            return np.full(8, float(self.ep), np.float64)

        def step(self, a):
            self.k += 1
            d = self.k >= self.lens[self.ep % len(self.lens)]
            ev.append(["env_step", [float(x) for x in np.asarray(a).ravel()]])
            return np.full(8, self.ep + 0.01 * self.k, np.float64), 1.5 * self.k, d, {}

    class FakeAgent:
        def __init__(self, *a, **k):
            pass

        def get_weights(self):
            return ["main/pi/w"], [np.zeros(1, np.float32)]

        def set_weights(self, keys, w):
            ev.append(["set_weights", list(keys)])

        def get_action(self, o, deterministic=False):
            ev.append(["get_action", float(np.asarray(o).ravel()[0])])
            return np.array([0.1, 0.2], np.float32)

        n_train = 0

        def train(self, *a):
            FakeAgent.n_train += 1
            if FakeAgent.n_train > 601:
                raise _Stop()

    class Remote:
        def __init__(self, fn):
            self.remote = fn

    class FakePS:
        def __init__(self):
            self.pull = Remote(lambda keys: (ev.append(["pull", list(keys)]), [np.zeros(1, np.float32)])[1])
            self.push = Remote(lambda keys, vals: ev.append(["push", FakeAgent.n_train]))

    class FakeRB:
        def __init__(self, counts=None):
            self.counts = counts
            self.store = Remote(lambda o, a, r, o2, d: ev.append(
                ["store", float(np.asarray(o).ravel()[0]), [float(x) for x in np.asarray(a).ravel()],
                 float(r), float(np.asarray(o2).ravel()[0]), bool(d)]))
            self.get_counts = Remote(self._counts)

        def _counts(self):
            c = self.counts.pop(0) if len(self.counts) > 1 else self.counts[0]
            ev.append(["get_counts", list(c)])
            return tuple(c)

    class Args:
        pass

    out = {}
    # ---- example/dsac.py worker_rollout (dsac.py:76-130)
    args = Args()
    args.env, args.steps_per_epoch, args.epochs, args.start_steps, args.max_ep_len = "fake", 14, 1, 4, 5
    dsac.gym.make = lambda name: FakeEnv()
    dsac.Model = FakeAgent
    dsac.worker_rollout(FakePS(), FakeRB(), args)
    out["dsac_rollout"] = {"args": {"total_steps": 14, "start_steps": 4, "max_ep_len": 5,
                                    "episode_lens": FakeEnv.lens}, "events": list(ev)}

    # ---- example/dsac.py worker_train (dsac.py:133-150): 601 iterations -> pushes at 300, 600
    ev.clear()
    FakeAgent.n_train = 0
    try:
        dsac.worker_train(FakePS(), FakeRB(), args)
    except _Stop:
        pass
    out["dsac_train"] = {"iterations": 601, "events": [e for e in ev if e[0] in ("push", "pull", "set_weights")]}

    # ---- algos/sac1/sac1.py worker_rollout (sac1.py:157-213): throttle on a_l_ratio
    ev.clear()
    opt = Args()
    opt.env_name, opt.obs_noise, opt.act_noise, opt.reward_scale = "fake", 0, 0, 1
    opt.steps_per_epoch, opt.total_epochs, opt.start_steps, opt.max_ep_len, opt.a_l_ratio = 5, 1, 2, 5, 2
    sac1.Wrapper = lambda env, *a: env
    sac1.gym.make = lambda name: FakeEnv()
    sac1.Actor = FakeAgent
    sac1.time.sleep = lambda s: ev.append(["sleep", s])
    # counts fed to the throttle: (sample_times, steps, size). 1st episode end: 0 samples -> no spin;
    # 2nd: steps/sample_times = 3 > 2 -> spin twice until ratio 2.0; afterwards raise to stop
    counts = [[0, 3, 3], [2, 6, 6], [2, 6, 6], [3, 6, 6], [1, 1, 1]]

    class StopRB(FakeRB):
        def _counts(self):
            if len(self.counts) == 1:
                raise _Stop()
            return super()._counts()
    try:
        sac1.worker_rollout(FakePS(), StopRB(counts), opt, 0)
    except _Stop:
        pass
    out["sac1_rollout"] = {"args": {"start_steps": 2, "max_ep_len": 5, "a_l_ratio": 2,
                                    "episode_lens": FakeEnv.lens,
                                    "counts_script": [[0, 3, 3], [2, 6, 6], [2, 6, 6], [3, 6, 6]]},
                           "events": list(ev)}
    with open(os.path.join(OUT, "worker_traces.json"), "w") as f:
        json.dump(out, f, indent=0)



class _NpProxy:
    """numpy as algos/dqn/train.py sees it, with np.random.seed() / np.random.choice(n, 1) scripted and recorded (the reference
    reseeds from the OS before every buffer choice: the choice is random BY DESIGN — the trace pins what is done with it)."""

    def __init__(self, ev, choices):
        self._ev, self._choices = ev, list(choices)
        self.random = self

    def seed(self, *a):
        self._ev.append(["np_seed"])

    def choice(self, n, k):
        v = self._choices.pop(0) if self._choices else 0
        self._ev.append(["choice", int(n), int(v)])
        return np.array([v])

    def __getattr__(self, name):
        return getattr(np, name)


def gen_dqn_driver_traces(dqn):
    """algos/dqn/train.py: worker_rollout (234-287), worker_train (213-231), Cache.ps_update (188-205), worker_test + get_al_status
    (289-371) — the event order of each, recorded by running the reference's own functions on scripted fakes."""
    ev = []

    class Remote:
        def __init__(self, fn):
            self.remote = fn

    class Args:
        pass

    class FakeSpace:
        def sample(self):
            ev.append(["sample_random"])
            return 2

    class FakeEnv:
        lens = [3, 2, 4]
        rewards = [7.5]

        def __init__(self, *a, **k):
            self.action_space = FakeSpace()
            self.ep, self.k = -1, 0

        def reset(self):
            self.ep += 1
            if self.ep >= len(self.lens):
                raise _Stop()
            self.k = 0
            ev.append(["reset"])
            return np.full(4, float(self.ep), np.float64)

        def step(self, a):
            self.k += 1
            ev.append(["env_step", int(a)])
            return np.full(4, self.ep + 0.01 * self.k, np.float64), 0.5 * self.k, self.k >= self.lens[self.ep], {}

    class FakeAgent:
        n_train = 0

        def __init__(self, opt, job):
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
            self.pull = Remote(lambda keys: (ev.append(["pull", list(keys)]), [np.zeros(1, np.float32)])[1])
            self.push = Remote(lambda keys, vals: ev.append(["push", idx, list(keys)]))
            self.get_weights = Remote(lambda: (ev.append(["ps_get_weights"]), {"main/q/w": np.zeros(1, np.float32), "target/q/w": np.ones(1, np.float32)})[1])
            self.save_weights = Remote(lambda: (ev.append(["ps_save_weights", idx]), "psop")[1])

    class FakeRB:
        def __init__(self, name, counts):
            self.name, self.counts = name, list(counts)
            self.store = Remote(lambda o, a, r, o2, d, wi: ev.append(
                ["store", name, float(np.asarray(o).ravel()[0]), int(a), float(r), float(np.asarray(o2).ravel()[0]), bool(d), int(wi)]))
            self.get_counts = Remote(self._counts)
            self.sample_batch = Remote(lambda: (ev.append(["sample_batch", name]), {"id": 0})[1])
            self.save = Remote(lambda: (ev.append(["rb_save", name]), "rbop")[1])

        def _counts(self):
            c = self.counts.pop(0) if len(self.counts) > 1 else self.counts[0]
            ev.append(["get_counts", self.name, list(c)])
            return tuple(c)

    out = {}
    fake_env_mod = types.ModuleType("trading_env")
    fake_env_mod.TradingEnv, fake_env_mod.FrameStack = FakeEnv, None
    sys.modules["trading_env"] = fake_env_mod
    real_np = dqn.np
    # ---- worker_rollout: two buffers; episode 1 below start_steps (random actions), episodes 2-3 above it
    opt = Args()
    opt.num_buffers, opt.start_steps, opt.recover = 2, 10, False
    choices = [1, 0, 1, 1, 0, 0, 1, 0, 1, 1, 0, 1]
    dqn.np = _NpProxy(ev, choices)
    dqn.Actor = FakeAgent
    rbs = [FakeRB("b0", [[0, 4, 4], [5, 9, 9]]), FakeRB("b1", [[0, 3, 3], [2, 6, 6], [4, 8, 8]])]
    try:
        dqn.worker_rollout(FakePS(), rbs, opt, 3)
    except _Stop:
        pass
    out["rollout"] = {"args": {"num_buffers": 2, "start_steps": 10, "recover": False, "worker_index": 3, "episode_lens": FakeEnv.lens,
                               "choices": choices, "counts": {"b0": [[0, 4, 4], [5, 9, 9]], "b1": [[0, 3, 3], [2, 6, 6], [4, 8, 8]]}},
                      "events": list(ev)}
    # ---- the same with opt.recover: the policy acts from the first step whatever the counters say
    ev.clear()
    opt.recover = True
    dqn.np = _NpProxy(ev, [0, 0, 0, 0])
    FakeEnv.lens = [2]
    try:
        dqn.worker_rollout(FakePS(), [FakeRB("b0", [[0, 0, 0]])], opt, 0)
    except _Stop:
        pass
    out["rollout_recover"] = {"events": list(ev)}
    FakeEnv.lens = [3, 2, 4]
    dqn.np = real_np
    # ---- worker_train: pull, set, cache.start(), then train(batch, cnt) on what q1 hands out, weights into q2 every push_freq
    ev.clear()

    class FakeQueue:
        def __init__(self, name):
            self.name, self.n = name, 0

        def get(self):
            self.n += 1
            return {"id": self.n}

        def put(self, x):
            ev.append(["%s_put" % self.name, list(x[0])])

    class FakeCache:
        def __init__(self, node_buffer):
            ev.append(["cache_init"])
            self.q1, self.q2 = FakeQueue("q1"), FakeQueue("q2")

        def start(self):
            ev.append(["cache_start"])

    opt = Args()
    opt.push_freq = 3
    real_cache = dqn.Cache
    dqn.Learner, dqn.Cache = FakeAgent, FakeCache
    try:
        dqn.worker_train(FakePS(), [[None]], opt, 0)
    except _Stop:
        pass
    out["train"] = {"args": {"push_freq": 3, "stop_at_cnt": 7}, "events": list(ev)}
    # ---- Cache.ps_update (the helper process): one batch up front, then a batch whenever fewer than 10 wait, pushes to EVERY node's server
    ev.clear()
    opt = Args()
    opt.num_nodes, opt.num_buffers = 2, 2
    dqn.opt = opt
    dqn.node_ps = [FakePS(0), FakePS(1)]
    choices = [1, 0, 0, 1, 1, 1, 0, 0]
    dqn.np = _NpProxy(ev, choices)
    sizes, empties = [1, 10, 12, 9, 10], [True, False, True, True, False]

    class ScriptQ1:
        def qsize(self):
            if not sizes:
                raise _Stop()
            v = sizes.pop(0)
            ev.append(["q1_qsize", v])
            return v

        def put(self, b):
            ev.append(["q1_put"])

    class ScriptQ2:
        def empty(self):
            v = empties.pop(0)
            ev.append(["q2_empty", v])
            return v

        def get(self):
            ev.append(["q2_get"])
            return ["main/q/w"], [np.zeros(1, np.float32)]

    nb = [[FakeRB("n0b0", [[0, 0, 0]]), FakeRB("n0b1", [[0, 0, 0]])], [FakeRB("n1b0", [[0, 0, 0]]), FakeRB("n1b1", [[0, 0, 0]])]]
    try:
        real_cache.ps_update(None, ScriptQ1(), ScriptQ2(), nb)
    except _Stop:
        pass
    dqn.np = real_np
    out["cache"] = {"args": {"num_nodes": 2, "num_buffers": 2, "choices": choices, "q1_sizes": [1, 10, 12, 9, 10],
                             "q2_empty": [True, False, True, True, False]}, "events": list(ev)}
    # ---- worker_test: two rounds; scripted counters and clock
    ev.clear()
    import tempfile
    tmp = tempfile.mkdtemp()
    opt = Args()
    opt.num_nodes, opt.num_buffers, opt.save_interval, opt.checkpoint_freq, opt.save_dir = 1, 2, 1000, 100.0, tmp
    dqn.opt = opt
    dqn.node_ps = [FakePS(0)]
    results = [(1.5, 7.5), (2.5, 8.5)]

    class FakeTester(FakeAgent):
        def test(self, env, n):
            if not results:
                raise _Stop()
            ev.append(["agent_test", int(n)])
            return results.pop(0)

        def write_tb(self, ave_test_reward, ave_score, alratio, update_frequency, total_learner_step):
            ev.append(["write_tb", float(ave_test_reward), float(ave_score), float(alratio), int(update_frequency), int(total_learner_step)])

    clock = [0.0, 10.0, 12.0, 12.0, 12.0, 50.0, 60.0, 62.0, 130.0, 130.0, 131.0]

    class TimeProxy:
        @staticmethod
        def time():
            v = clock.pop(0) if len(clock) > 1 else clock[0]
            ev.append(["time", v])
            return v

    real_time, real_wait = dqn.time, dqn.ray.wait
    dqn.time = TimeProxy
    dqn.ray.wait = lambda ops, num_returns=1: (ev.append(["ray_wait", len(ops), int(num_returns)]), (ops, []))[1]
    dqn.ray.cluster_resources = lambda: {}
    dqn.ray.available_resources = lambda: {}
    dqn.Actor = FakeTester
    # (learner_steps, actor_steps, size) per buffer and call: round 1: 40 -> 1250 learner steps over both buffers; round 2: -> 2600
    nb = [[FakeRB("b0", [[20, 100, 100], [600, 300, 300], [600, 300, 300], [1300, 500, 500]]),
           FakeRB("b1", [[20, 100, 100], [650, 320, 320], [650, 320, 320], [1300, 520, 520]])]]
    import builtins
    real_print = builtins.print
    builtins.print = lambda *a, **k: None
    try:
        dqn.worker_test(FakePS(0), nb, opt)
    except _Stop:
        pass
    finally:
        builtins.print = real_print
        dqn.time, dqn.ray.wait = real_time, real_wait
    saved = sorted(os.listdir(tmp))
    out["test"] = {"args": {"num_nodes": 1, "num_buffers": 2, "save_interval": 1000, "checkpoint_freq": 100.0,
                            "clock": [0.0, 10.0, 12.0, 12.0, 12.0, 50.0, 60.0, 62.0, 130.0, 130.0, 131.0],
                            "results": [[1.5, 7.5], [2.5, 8.5]]},
                   "events": list(ev), "files": saved}
    with open(os.path.join(OUT, "dqn_driver_traces.json"), "w") as f:
        json.dump(out, f, indent=0)
    sys.modules.pop("trading_env", None)



def gen_nstep_driver_traces(sac_ray):
    """algos/sac1/sac_ray.py: worker_rollout (178-274: the window deques, the random-action budget, store cadence, the pull that waits
    for the buffer's start_steps), worker_train (155-175: push every 100 through the Cache), Cache.ps_update (135-147)."""
    ev = []

    class Remote:
        def __init__(self, fn):
            self.remote = fn

    class Args:
        pass

    class FakeSpace:
        def sample(self):
            ev.append(["sample_random"])
            return np.array([0.5, -0.5], np.float32)

    class FakeEnv:
        lens = [6, 99, 3]

        def __init__(self, *a, **k):
            self.action_space = FakeSpace()
            self.ep, self.k = -1, 0

        def reset(self):
            self.ep += 1
            if self.ep >= len(self.lens):
                raise _Stop()
            self.k = 0
            ev.append(["reset"])
            return np.full(3, float(self.ep), np.float64)

        def step(self, a):
            self.k += 1
            ev.append(["env_step", [float(x) for x in np.asarray(a).ravel()]])
            return np.full(3, self.ep + 0.01 * self.k, np.float64), 0.25 * self.k, self.k >= self.lens[self.ep], {}

    class FakeAgent:
        def __init__(self, opt, job):
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
        def __init__(self):
            self.pull = Remote(lambda keys: (ev.append(["pull", list(keys)]), [np.zeros(1, np.float32)])[1])
            self.push = Remote(lambda keys, vals: ev.append(["push", list(keys)]))

    class FakeRB:
        def __init__(self, name, counts):
            self.name, self.counts = name, [list(c) for c in counts]
            self.store = Remote(lambda oq, aq, wi: ev.append(
                ["store", name, [float(np.asarray(q[0]).ravel()[0]) for q in oq], [[float(q[0][0]), float(q[1]), bool(q[2])] for q in aq], int(wi)]))
            self.get_counts = Remote(self._counts)
            self.sample_batch = Remote(lambda: (ev.append(["sample_batch", name]), {"id": 0})[1])

        def _counts(self):
            c = self.counts.pop(0) if len(self.counts) > 1 else self.counts[0]
            ev.append(["get_counts", self.name, list(c)])
            return tuple(c)

    out = {}
    real_np = sac_ray.np
    opt = Args()
    opt.env_name, opt.obs_noise, opt.act_noise, opt.reward_scale, opt.model = "fake", 0, 0, 1, "mlp"
    opt.Ln, opt.save_freq, opt.num_buffers, opt.start_steps, opt.weights_file, opt.action_repeat, opt.max_ep_len = 3, 2, 2, 4, "", 2, 10
    sac_ray.Wrapper = lambda env, *a: env
    sac_ray.gym.make = lambda name: FakeEnv()
    sac_ray.Actor = FakeAgent
    choices = [1, 0, 1, 1, 0]
    sac_ray.np = _NpProxy(ev, choices)
    counts0 = [[0, 3, 3], [1, 9, 9], [2, 12, 12]]
    import builtins
    real_print = builtins.print
    builtins.print = lambda *a, **k: None
    try:
        sac_ray.worker_rollout(FakePS(), [FakeRB("b0", counts0), FakeRB("b1", [[0, 0, 0]])], opt, 5)
    except _Stop:
        pass
    finally:
        builtins.print = real_print
    out["rollout"] = {"args": {"Ln": 3, "save_freq": 2, "num_buffers": 2, "start_steps": 4, "action_repeat": 2, "max_ep_len": 10, "worker_index": 5,
                               "episode_lens": FakeEnv.lens, "choices": choices, "counts_b0": counts0}, "events": list(ev)}
    # ---- worker_train: push every 100 updates through the Cache's q2
    ev.clear()
    sac_ray.np = real_np

    class FakeQueue:
        def __init__(self, name):
            self.name, self.n = name, 0

        def get(self):
            self.n += 1
            return {"id": self.n}

        def put(self, x):
            ev.append(["%s_put" % self.name, list(x[0])])

    class FakeCache:
        def __init__(self, replay_buffer):
            ev.append(["cache_init"])
            self.q1, self.q2 = FakeQueue("q1"), FakeQueue("q2")

        def start(self):
            ev.append(["cache_start"])

    real_cache = sac_ray.Cache
    sac_ray.Learner, sac_ray.Cache = FakeAgent, FakeCache
    try:
        sac_ray.worker_train(FakePS(), [None], opt, 0)
    except _Stop:
        pass
    out["train"] = {"events": list(ev)}
    # ---- Cache.ps_update: the single-server form (global `ps`, global `opt`)
    ev.clear()
    sac_ray.opt = opt
    sac_ray.ps = FakePS()
    choices = [1, 0, 1]
    sac_ray.np = _NpProxy(ev, choices)
    sizes, empties = [2, 10, 9], [False, True, True]

    class ScriptQ1:
        def qsize(self):
            if not sizes:
                raise _Stop()
            v = sizes.pop(0)
            ev.append(["q1_qsize", v])
            return v

        def put(self, b):
            ev.append(["q1_put"])

    class ScriptQ2:
        def empty(self):
            v = empties.pop(0)
            ev.append(["q2_empty", v])
            return v

        def get(self):
            ev.append(["q2_get"])
            return ["main/pi/w"], [np.zeros(1, np.float32)]

    builtins.print = lambda *a, **k: None
    try:
        real_cache.ps_update(None, ScriptQ1(), ScriptQ2(), [FakeRB("b0", [[0, 0, 0]]), FakeRB("b1", [[0, 0, 0]])])
    except _Stop:
        pass
    finally:
        builtins.print = real_print
    sac_ray.np = real_np
    out["cache"] = {"args": {"num_buffers": 2, "choices": choices, "q1_sizes": [2, 10, 9], "q2_empty": [False, True, True]}, "events": list(ev)}
    with open(os.path.join(OUT, "nstep_driver_traces.json"), "w") as f:
        json.dump(out, f, indent=0)


def nstep_windows(n_store, Ln=4, obs_dim=115):
    """The deque contents a sac_ray-style rollout hands to store(): yields (o_queue, a_r_d_queue)
    snapshots (deques of (o,) tuples / (a, r, d) tuples), scalar actions (act_shape == ())."""
    from collections import deque
    o_queue, a_r_d_queue = deque([], maxlen=Ln + 1), deque([], maxlen=Ln)
    rs = np.random.RandomState(n_store)
    o_queue.append((rs.randn(obs_dim),))
    t, stored = 1, 0
    while stored < n_store:
        a, r, d = float(rs.uniform(-1, 1)), float(rs.randn()), bool(rs.rand() < 0.2)
        a_r_d_queue.append((a, r, d,))
        o_queue.append((rs.randn(obs_dim),))
        if t >= Ln:
            yield deque(o_queue, maxlen=Ln + 1), deque(a_r_d_queue, maxlen=Ln)
            stored += 1
        t += 1


def gen_nstep(sac_ray):
    """n-step window buffer of algos/sac1/sac_ray.py:34-82: (Ln+1) observation frames and Ln
    (action, reward, done) triples per slot; counters advance by num_buffers."""
    class Opt:
        pass
    opt = Opt()
    opt.obs_shape, opt.act_shape, opt.Ln, opt.buffer_size, opt.batch_size, opt.num_buffers = (115,), (), 4, 7, 5, 3
    rec, meta = {}, []
    for n_store in (5, 7, 10):
        buf = sac_ray.ReplayBuffer(opt)
        for oq, aq in nstep_windows(n_store, opt.Ln, 115):
            buf.store(oq, aq, 0)
        tag = "nstep_n%d" % n_store
        for k in ("buffer_o", "buffer_a", "buffer_r", "buffer_d"):
            rec[tag + "_" + k] = getattr(buf, k).copy()
        np.random.seed(21)
        batches = []
        for it in range(2):
            dct = buf.sample_batch()
            for k, v in dct.items():
                rec["%s_s%d_%s" % (tag, it, k)] = v
            batches.append(_mt_fingerprint())
        meta.append({"n_store": n_store, "ptr": int(buf.ptr), "size": int(buf.size), "counts": _jsonable(buf.get_counts()),
                     "Ln": opt.Ln, "obs": 115, "cap": opt.buffer_size, "B": opt.batch_size, "num_buffers": opt.num_buffers,
                     "seed": 21, "mt": batches})
    np.savez_compressed(os.path.join(OUT, "nstep.npz"), **rec)
    with open(os.path.join(OUT, "nstep.json"), "w") as f:
        json.dump(meta, f, indent=0)


def main():
    if not os.path.isdir(REF):
        raise SystemExit("reference checkout not present; fixtures can only be regenerated in the build container")
    os.makedirs(OUT, exist_ok=True)
    _install_stubs()
    dsac = _load(os.path.join(REF, "example", "dsac.py"), "ref_dsac", os.path.join(REF, "example"))
    sys.path.remove(os.path.join(REF, "example"))
    for m in ("core", "model"):
        sys.modules.pop(m, None)
    sac1 = _load(os.path.join(REF, "algos", "sac1", "sac1.py"), "ref_sac1", os.path.join(REF, "algos", "sac1"))
    sys.path.remove(os.path.join(REF, "algos", "sac1"))
    for m in ("core", "hyperparams", "actor_learner"):
        sys.modules.pop(m, None)
    sys.path.insert(0, os.path.join(REF, "algos"))
    dqn = _load(os.path.join(REF, "algos", "dqn", "train.py"), "ref_dqn", os.path.join(REF, "algos", "dqn"))
    gen_index_streams(dsac)
    gen_ring_and_gather(dsac, sac1, dqn)
    gen_ps(dsac)
    gen_worker_traces(dsac, sac1)
    gen_dqn_driver_traces(dqn)
    sys.path.remove(os.path.join(REF, "algos", "dqn"))
    for m in ("core", "hyperparams", "actor_learner", "trading_env"):
        sys.modules.pop(m, None)
    sac_ray = _load(os.path.join(REF, "algos", "sac1", "sac_ray.py"), "ref_sac_ray", os.path.join(REF, "algos", "sac1"))
    gen_nstep(sac_ray)
    gen_nstep_driver_traces(sac_ray)
    import subprocess
    subprocess.check_call([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "gen_golden_math.py")])
    print("golden fixtures written to", OUT)
    for fn in sorted(os.listdir(OUT)):
        print("  %-28s %8d B" % (fn, os.path.getsize(os.path.join(OUT, fn))))


if __name__ == "__main__":
    main()
