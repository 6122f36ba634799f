"""Host logic (no GPU): worker control flow against traces recorded from the reference's OWN
worker_rollout / worker_train functions (tests/golden/worker_traces.json, oracle/gen_golden.py),
the Ray-shaped remote shim, and the C-ABI library's exports."""
import json
import os
import re
import threading

import numpy as np
import pytest

from distributed_drl_amd import remote as ray
from distributed_drl_amd import workers


class _Stop(Exception):
    pass


def _harness(ev, lens):
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
            self.k = 0
            ev.append(["reset"])
            return np.full(8, float(self.ep), np.float64)

        def step(self, a):
            self.k += 1
            d = self.k >= lens[self.ep % len(lens)]
            ev.append(["env_step", [float(x) for x in np.asarray(a).ravel()]])
            return np.full(8, self.ep + 0.01 * self.k, np.float64), 1.5 * self.k, d, {}

    class FakeAgent:
        n_train = 0

        def __init__(self, *a, **k):
            pass

        def get_weights(self):
            return ["main/pi/w"], [np.zeros(1, np.float32)]

        def set_weights(self, keys, w):
            ev.append(["set_weights", list(keys)])

        def get_action(self, o, deterministic=False):
            ev.append(["get_action", float(np.asarray(o).ravel()[0])])
            return np.array([0.1, 0.2], np.float32)

        def train(self, *a):
            FakeAgent.n_train += 1
            if FakeAgent.n_train > 601:
                raise _Stop()

    class FakePS:  # plain object: workers._remote falls back to a direct call
        def pull(self, keys):
            ev.append(["pull", list(keys)])
            return [np.zeros(1, np.float32)]

        def push(self, keys, vals):
            ev.append(["push", FakeAgent.n_train])

    class FakeRB:
        def __init__(self, counts=None):
            self.counts = counts

        def store(self, o, a, r, o2, d):
            ev.append(["store", float(np.asarray(o).ravel()[0]), [float(x) for x in np.asarray(a).ravel()],
                       float(r), float(np.asarray(o2).ravel()[0]), bool(d)])

        def get_counts(self):
            if len(self.counts) == 1:
                raise _Stop()
            c = self.counts.pop(0)
            ev.append(["get_counts", list(c)])
            return tuple(c)

    return FakeEnv, FakeAgent, FakePS, FakeRB


class _Args:
    pass


@pytest.fixture(scope="module")
def traces(golden_dir):
    return json.load(open(os.path.join(golden_dir, "worker_traces.json")))


def test_worker_rollout_event_order_matches_reference(traces):
    t = traces["dsac_rollout"]
    ev = []
    Env, Agent, PS, RB = _harness(ev, t["args"]["episode_lens"])
    args = _Args()
    args.env, args.steps_per_epoch, args.epochs = "fake", t["args"]["total_steps"], 1
    args.start_steps, args.max_ep_len = t["args"]["start_steps"], t["args"]["max_ep_len"]
    workers.worker_rollout(PS(), RB(), args, make_env=lambda n: Env(), make_agent=lambda a: Agent())
    assert ev == t["events"]
    # strict '>' : start_steps + 1 random actions
    assert sum(1 for e in ev if e[0] == "sample_random") == t["args"]["start_steps"] + 1
    # the time-limit transition is stored with done == False but ends the episode
    stores = [e for e in ev if e[0] == "store"]
    assert stores[7][5] is False and ev[ev.index(stores[7]) + 1] == ["reset"]


def test_worker_train_push_cadence_matches_reference(traces):
    t = traces["dsac_train"]
    ev = []
    Env, Agent, PS, RB = _harness(ev, [3])
    Agent.n_train = 0
    args = _Args()
    with pytest.raises(_Stop):
        workers.worker_train(PS(), RB(), args, make_agent=lambda a: Agent())
    assert [e for e in ev if e[0] in ("push", "pull", "set_weights")] == t["events"]


def test_worker_rollout_sac1_throttle_matches_reference(traces):
    t = traces["sac1_rollout"]
    ev = []
    Env, Agent, PS, RB = _harness(ev, t["args"]["episode_lens"])
    opt = _Args()
    opt.env_name, opt.start_steps, opt.max_ep_len = "fake", t["args"]["start_steps"], t["args"]["max_ep_len"]
    opt.a_l_ratio = t["args"]["a_l_ratio"]
    counts = [list(c) for c in t["args"]["counts_script"]] + [[1, 1, 1]]
    with pytest.raises(_Stop):
        workers.worker_rollout_sac1(PS(), RB(counts), opt, 0, make_env=lambda n: Env(), make_agent=lambda a: Agent(),
                                    sleep=lambda s: ev.append(["sleep", s]))
    assert ev == t["events"]


def test_remote_shim_actor_is_serial_and_tasks_are_concurrent():
    @ray.remote
    class Counter:
        def __init__(self, start):
            self.n = start
            self.active = 0
            self.overlap = False

        def inc(self):
            self.active += 1
            self.overlap |= self.active > 1
            n = self.n
            for _ in range(1000):
                pass
            self.n = n + 1
            self.active -= 1
            return self.n

        def boom(self):
            raise ValueError("high <= 0")

        def state(self):
            return self.n, self.overlap

    c = Counter.remote(10)

    @ray.remote(num_gpus=1, max_calls=1)
    def task(handle, k):
        return [ray.get(handle.inc.remote()) for _ in range(k)][-1]

    futs = [task.remote(c, 50) for _ in range(4)]
    ready, rest = ray.wait(futs, num_returns=4)
    assert len(ready) == 4 and not rest
    n, overlap = ray.get(c.state.remote())
    assert n == 210 and overlap is False          # serial execution inside one actor
    with pytest.raises(ValueError):               # method errors surface at get()
        ray.get(c.boom.remote())
    assert ray.get([c.inc.remote(), c.inc.remote()]) == [211, 212]   # arrival order
    assert ray.get(5) == 5 and ray.init() is None


def test_dsac_shaped_driver_runs_through_the_shim(traces):
    """A driver with the structure of example/dsac.py:218-238 (actors + tasks + ray.wait)."""
    ev = []
    lock = threading.Lock()

    class SafeList(list):
        def append(self, x):
            with lock:
                list.append(self, x)
    ev = SafeList()
    Env, Agent, PS, RB = _harness(ev, [3, 99, 2, 99])
    args = _Args()
    args.env, args.steps_per_epoch, args.epochs, args.start_steps, args.max_ep_len = "fake", 14, 1, 4, 5
    ps = ray.remote(PS).remote()
    rb = ray.remote(RB).remote()
    roll = ray.remote(workers.worker_rollout)
    tasks = [roll.remote(ps, rb, args, lambda n: Env(), lambda a: Agent()) for _ in range(2)]
    ready, _ = ray.wait(tasks, num_returns=2)
    ray.get(ready)
    # store.remote() is fire-and-forget (dsac.py:112): drain the actor's FIFO mailbox with one awaited call
    ray.get(rb.store.remote(np.zeros(8), np.zeros(2), 0.0, np.zeros(8), False))
    assert sum(1 for e in ev if e[0] == "store") == 28 + 1


def test_library_exports_every_declared_symbol():
    """include/ddrl.h <-> ctypes table <-> libddrl_hip.so (loads on CPU; no compute call)."""
    from distributed_drl_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "ddrl.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(ddrl_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    lib = _lib.load()  # raises if the .so is missing or lacks a symbol
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.ddrl_version() == 100
    n_pi, n_q = _lib.c_int64(), _lib.c_int64()
    cfg = _lib.Sac1Config()
    import ctypes
    _lib.check(lib.ddrl_sac1_param_counts(ctypes.byref(cfg), ctypes.byref(n_pi), ctypes.byref(n_q)))
    assert (n_pi.value, n_q.value) == (125104, 125001)
    with pytest.raises(ValueError):
        cfg.act_dim = 99
        _lib.check(lib.ddrl_sac1_param_counts(ctypes.byref(cfg), None, None))


def test_product_path_does_not_import_the_oracle():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.join(root, "distributed-drl_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, fn)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), fn
                assert "liboracle" not in src, fn


def test_agent_param_specs_match_oracle_layout():
    from distributed_drl_amd.agent import glorot_init, param_specs
    from oracle import sac1_oracle as so
    cfg = so.Config()
    assert param_specs(8, 2, 400, 300) == so.param_specs(cfg)
    np.testing.assert_array_equal(glorot_init(param_specs(8, 2, 400, 300), 7), so.flatten(so.init_params(cfg, 7)))


def test_epoch_logger_progress_file(tmp_path):
    """The logger slice example/dsac.py:153-177 uses: config.json + tab-separated progress.txt with a header."""
    from distributed_drl_amd.logx import EpochLogger, setup_logger_kwargs
    kw = setup_logger_kwargs("dsac", 3, data_dir=str(tmp_path))
    assert kw["output_dir"].endswith("dsac/dsac_s3")
    lg = EpochLogger(quiet=True, **kw)
    lg.save_config(dict(seed=3, env="LunarLanderContinuous-v2", fn=print))
    for i in range(3):
        lg.log_tabular("AverageTestEpRet", -100.0 + i)
        lg.log_tabular("Time", 1.5 * i)
        lg.dump_tabular()
    rows = open(tmp_path / "dsac" / "dsac_s3" / "progress.txt").read().strip().split("\n")
    assert rows[0] == "AverageTestEpRet\tTime" and rows[3] == "-98.0\t3.0" and len(rows) == 4
    import json
    assert json.load(open(tmp_path / "dsac" / "dsac_s3" / "config.json"))["exp_name"] == "dsac"
    with pytest.raises(AssertionError):
        lg.log_tabular("NewKey", 1)


def test_tensorboard_event_file_roundtrip(tmp_path):
    """The scalar stream Actor.test() writes in place of tf.summary.FileWriter (actor_learner.py:210-229): TFRecord framing with
    masked crc32c (known answer: crc32c(b"123456789") = 0xE3069283), hand-encoded Event / Summary protobufs, read back."""
    from distributed_drl_amd import logx
    assert logx._crc32c(b"123456789") == 0xE3069283
    w = logx.SummaryWriter(str(tmp_path))
    for step, v in ((0, -183.5), (300, 12.25), (70000, 251.0)):
        w.add_scalar("Reward", v, step)
    w.close()
    assert logx.read_scalars(w.path) == [(0, "Reward", -183.5), (300, "Reward", 12.25), (70000, "Reward", 251.0)]
    raw = open(w.path, "rb").read()
    assert b"brain.Event:2" in raw[:64] and os.path.basename(w.path).startswith("events.out.tfevents.")


def test_traffic_file_was_taken_on_the_current_kernels():
    """profiles/traffic.json (the PMC passes behind bench.py's roofline.traffic) carries the sha256 of csrc/*.h, *.hip it was taken on:
    a kernel change without a new tools/prof_round.sh + tools/make_traffic.py pass fails here, not silently in the bench line."""
    import json
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    t = json.load(open(os.path.join(root, "profiles", "traffic.json")))
    assert t.get("kernel_source_sha256") == bench.kernel_source_hash()
