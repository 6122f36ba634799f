#!/usr/bin/env python3
"""Generate the learner-math golden fixtures by EXECUTING the reference's own learner / actor text.

TEST INFRASTRUCTURE ONLY; runs only in the build container (reads /root/reference).  TensorFlow,
Ray, gym and spinup are absent from this image, so `tensorflow` and `ray.experimental.tf_utils` are
replaced by oracle/tf_shim.py (a lazy graph over torch float64 — its header lists exactly what is
"from memory" there: dense = x@W+b and its variable naming, ApplyAdam, make_template, the variable
collection, sess.run ordering).  With that in sys.modules this script imports and runs

    algos/sac1/actor_learner.py   Learner.__init__ / set_weights / get_weights / train, Actor.get_action
    algos/sac1/core.py            mlp_actor_critic and everything under it
    example/model.py, core.py     Model.__init__ / set_weights / train-step fetches / get_action
    algos/dqn/actor_learner.py    Learner (Double-DQN), core.q_function
    algos/sqn/actor_learner.py    Learner (SQN), core.q_function / softmax_policy

feeds them the seeded inputs of oracle/fixture_inputs.py and writes what they return — losses,
per-row outputs, the gradient each optimizer was handed per variable, parameters / targets / Adam
slots after each of several sequential train() calls, the variable names and order of
get_weights() — to tests/golden/{sac1,sacv,dqn,sqn}_math.{npz,json}.  Large tensors are kept as
digests (fixture_inputs.digest).  No reference source is copied; only vectors are written.

Usage:  python oracle/gen_golden_math.py [family ...]
"""
import importlib.util
import json
import os
import sys
import types
from unittest import mock

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import fixture_inputs as fi   # noqa: E402
from oracle import tf_shim                # noqa: E402

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def _install():
    tf, tfu = tf_shim.as_modules()
    ray = types.ModuleType("ray")
    ray.remote = lambda *a, **k: (a[0] if len(a) == 1 and not k and callable(a[0]) else (lambda o: o))
    ray.get = lambda x: x
    ray.experimental = types.ModuleType("ray.experimental")
    ray.experimental.tf_utils = tfu
    sys.modules.update({"tensorflow": tf, "ray": ray, "ray.experimental": ray.experimental,
                        "ray.experimental.tf_utils": tfu})
    for name in ("gym", "gym.spaces", "spinup", "spinup.algos", "spinup.algos.sac", "spinup.utils", "spinup.utils.logx",
                 "spinup.utils.run_utils"):
        sys.modules[name] = mock.MagicMock()
    return tf


def _load_from(directory, filename, modname, forget=("core", "hyperparams", "actor_learner", "model")):
    for m in forget:
        sys.modules.pop(m, None)
    sys.path.insert(0, directory)
    try:
        spec = importlib.util.spec_from_file_location(modname, os.path.join(directory, filename))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        sys.path.remove(directory)
    return mod


class _Opt:
    def __init__(self, **k):
        self.__dict__.update(k)


class _Space:
    def __init__(self, high, dim):
        self.high = np.full(dim, high, np.float32)
        self.shape = (dim,)


def _graph_of(agent):
    return agent.sess.graph


def _model_names(agent):
    keys, values = agent.get_weights()
    return list(keys), [tuple(np.asarray(v).shape) for v in values]


def _set_targets(graph, names, values):
    """Give the target network its own parameters (a state the reference reaches after training; right after
    set_weights target == main, which would hide a main/target mix-up from every later comparison)."""
    for n, v in zip(names, values):
        graph.by_name[n.replace("main/", "target/", 1)].assign_value(v)


def _state(graph, names, rec, tag):
    for i, n in enumerate(names):
        rec["%s_main_%d" % (tag, i)] = fi.digest(graph.by_name[n].value.numpy())
        rec["%s_targ_%d" % (tag, i)] = fi.digest(graph.by_name[n.replace("main/", "target/", 1)].value.numpy())
        for slot, key in (("Adam", "m"), ("Adam_1", "v")):
            rec["%s_%s_%d" % (tag, key, i)] = fi.digest(graph.by_name[n + "/" + slot].value.numpy())


def _grads(graph, names, rec, tag):
    got = {}
    for opt in graph.optimizers:
        for k, g in opt.last_grads.items():
            assert k not in got, "a variable stepped by two optimizers"
            got[k] = g
    assert set(got) == set(names), (sorted(set(names) ^ set(got)))
    for i, n in enumerate(names):
        rec["%s_grad_%d" % (tag, i)] = fi.digest(got[n].numpy())


def _save(family, rec, meta):
    np.savez_compressed(os.path.join(OUT, family + "_math.npz"), **rec)
    with open(os.path.join(OUT, family + "_math.json"), "w") as f:
        json.dump(meta, f, indent=1)


# ------------------------------------------------------------------------------------------------
def gen_sac1():
    al = _load_from(os.path.join(REF, "algos", "sac1"), "actor_learner.py", "ref_sac1_al")
    cases = [dict(tag="c0", obs_dim=8, act_dim=2, batch=256, act_high=1.0, alpha=0.1, gamma=0.997, lr=5e-5, polyak=0.995,
                  steps=3, seed=11),
             dict(tag="c1", obs_dim=3, act_dim=1, batch=37, act_high=2.0, alpha=0.2, gamma=0.99, lr=1e-3, polyak=0.9,
                  steps=2, seed=12)]
    rec, meta = {}, {"source": "algos/sac1/actor_learner.py Learner/Actor on oracle/tf_shim.py (float64)",
                     "noise_calls": ["main pi @ x (eps_x)", "main pi @ x2 (eps_x2)", "target pi @ x2 (eps_t)",
                                     "target pi @ x2, second policy() call of the target scope (never fetched)"],
                     "cases": []}
    for c in cases:
        opt = _Opt(seed=0, obs_dim=c["obs_dim"], act_dim=c["act_dim"], alpha=c["alpha"], gamma=c["gamma"], lr=c["lr"],
                   polyak=c["polyak"], gpu_fraction=0.3,
                   ac_kwargs={"action_space": _Space(c["act_high"], c["act_dim"])})
        learner = al.Learner(opt, "learner")
        g = _graph_of(learner)
        names, shapes = _model_names(learner)
        assert g.n_random == 4
        main = fi.make_params(list(zip(names, shapes)), c["seed"], "main")
        targ = fi.make_params(list(zip(names, shapes)), c["seed"], "target")
        learner.set_weights(names, main)
        # target_init ran: target == main (recorded), then the targets get their own values
        assert all((g.by_name[n.replace("main/", "target/", 1)].value.numpy() == m).all() for n, m in zip(names, main))
        _set_targets(g, names, targ)
        entry = dict(c, names=names, shapes=[list(s) for s in shapes],
                     all_variables=[v.op_name for v in g.variables],
                     optimizer_var_lists=[[v.op_name for v in o.var_list] for o in g.optimizers],
                     noise_used=[])
        for s in range(c["steps"]):
            batch, noise = fi.sac_batch(c["obs_dim"], c["act_dim"], c["batch"], 100 * c["seed"] + s, c["act_high"])
            learner.sess.noise = noise
            learner.train(batch)
            outs = learner.sess.last_outputs
            tag = "%s_s%d" % (c["tag"], s)
            for k, v in zip(("pi_loss", "q1_loss", "q2_loss", "q1", "q2", "logp_pi", "alpha"), outs[:7]):
                rec[tag + "_" + k] = np.asarray(v, np.float64)
            entry["noise_used"].append(learner.sess.last_noise_used)
            _grads(g, names, rec, tag)
            _state(g, names, rec, tag)
        # Actor: the reference's get_action, one observation per call, explicit noise
        actor = al.Actor(_Opt(seed=0, obs_dim=c["obs_dim"], act_dim=c["act_dim"], summary_dir="", env_name="", num_workers=1,
                              a_l_ratio=1, ac_kwargs=opt.ac_kwargs), "worker")
        a_names, _ = _model_names(actor)
        entry["actor_names"] = a_names
        actor.set_weights(names, main)            # the reference pushes the learner's whole dict; the actor keeps its own
        rs = np.random.RandomState(c["seed"])
        obs = rs.randn(9, c["obs_dim"]).astype(np.float32)
        eps = rs.randn(9, c["act_dim"]).astype(np.float32)
        acts, mus = [], []
        for i in range(9):
            actor.sess.noise = [eps[i:i + 1], None]
            acts.append(actor.get_action(obs[i]))
            mus.append(actor.get_action(obs[i], True))
        rec[c["tag"] + "_actor_obs"], rec[c["tag"] + "_actor_eps"] = obs, eps
        rec[c["tag"] + "_actor_pi"], rec[c["tag"] + "_actor_mu"] = np.asarray(acts, np.float64), np.asarray(mus, np.float64)
        meta["cases"].append(entry)
    _save("sac1", rec, meta)


def gen_sacv():
    ex = os.path.join(REF, "example")
    core = _load_from(ex, "core.py", "core", forget=("core",))
    # example/model.py imports placeholders / get_vars / count_vars from spinup's core (absent); example/core.py is
    # the reference's own copy of that file and defines the same three helpers
    sys.modules["spinup.algos.sac"].core = core
    sys.modules["spinup.algos.sac.core"] = core
    sys.path.insert(0, ex)
    try:
        sys.modules["core"] = core
        spec = importlib.util.spec_from_file_location("ref_model", os.path.join(ex, "model.py"))
        model = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(model)
    finally:
        sys.path.remove(ex)
    cases = [dict(tag="c0", obs_dim=8, act_dim=2, batch=100, hid=300, act_high=1.0, alpha=0.2, gamma=0.99, lr=1e-3,
                  polyak=0.995, steps=3, seed=21),
             dict(tag="c1", obs_dim=5, act_dim=3, batch=33, hid=64, act_high=1.5, alpha=0.1, gamma=0.9, lr=3e-4,
                  polyak=0.95, steps=2, seed=22)]
    rec, meta = {}, {"source": "example/model.py Model on oracle/tf_shim.py (float64); args as example/dsac.py:193-212 "
                               "(gamma is the 1-tuple `0.99,` there and is passed as such)", "cases": []}
    for c in cases:
        tf_shim.reset_default_graph()           # Model builds into the default graph
        args = _Opt(obs_dim=c["obs_dim"], act_dim=c["act_dim"], gamma=(c["gamma"],), alpha=c["alpha"], lr=c["lr"],
                    polyak=c["polyak"], batch_size=c["batch"],
                    ac_kwargs=dict(hidden_sizes=[c["hid"]] * 2, action_space=_Space(c["act_high"], c["act_dim"])))
        net = model.Model(args)
        g = _graph_of(net)
        names, shapes = _model_names(net)
        main = fi.make_params(list(zip(names, shapes)), c["seed"], "main")
        targ = fi.make_params(list(zip(names, shapes)), c["seed"], "target")
        net.set_weights(names, main)
        _set_targets(g, names, targ)
        entry = dict(c, names=names, shapes=[list(s) for s in shapes], n_random=g.n_random,
                     optimizer_var_lists=[[v.op_name for v in o.var_list] for o in g.optimizers], noise_used=[])
        for s in range(c["steps"]):
            batch, noise = fi.sac_batch(c["obs_dim"], c["act_dim"], c["batch"], 100 * c["seed"] + s, c["act_high"])
            net.sess.noise = noise[:g.n_random]

            class _RB:                            # Model.train(replay_buffer, args): batch = ray.get(rb.sample_batch.remote(B))
                class sample_batch:
                    remote = staticmethod(lambda n, _b=batch: _b)
            net.train(_RB, args)
            outs = net.sess.last_outputs
            tag = "%s_s%d" % (c["tag"], s)
            for k, v in zip(("pi_loss", "q1_loss", "q2_loss", "v_loss", "q1", "q2", "v", "logp_pi"), outs[:8]):
                rec[tag + "_" + k] = np.asarray(v, np.float64)
            entry["noise_used"].append(net.sess.last_noise_used)
            _grads(g, names, rec, tag)
            _state(g, names, rec, tag)
        meta["cases"].append(entry)
    _save("sacv", rec, meta)


def _gen_discrete(family, directory, cases, extra):
    al = _load_from(directory, "actor_learner.py", "ref_%s_al" % family)
    rec, meta = {}, {"source": "%s Learner on oracle/tf_shim.py (float64)" % os.path.relpath(directory, REF), "cases": []}
    for c in cases:
        opt = _Opt(seed=0, obs_dim=c["obs_dim"], act_dim=c["n_actions"], hidden_size=c["hidden"], gamma=c["gamma"],
                   lr=c["lr"], polyak=c["polyak"], gpu_fraction=0.3, summary_dir="", env_name="", exp_name="",
                   num_workers=1, a_l_ratio=1, **extra)
        learner = al.Learner(opt, "learner")
        g = _graph_of(learner)
        names, shapes = _model_names(learner)
        main = fi.make_params(list(zip(names, shapes)), c["seed"], "main")
        targ = fi.make_params(list(zip(names, shapes)), c["seed"], "target")
        if c.get("pixels"):                      # keep layer 1 of a 28 224-wide pixel input in range
            main[0] *= np.float32(1.0 / 64)
            targ[0] *= np.float32(1.0 / 64)
        learner.set_weights(names, main)
        _set_targets(g, names, targ)
        entry = dict(c, names=names, shapes=[list(s) for s in shapes],
                     optimizer_var_lists=[[v.op_name for v in o.var_list] for o in g.optimizers])
        for s in range(c["steps"]):
            batch = fi.dqn_batch(c["obs_dim"], c["n_actions"], c["batch"], 100 * c["seed"] + s, c.get("pixels", False))
            learner.train(batch, 1)               # cnt = 1: no summary branch
            outs = learner.sess.last_outputs
            tag = "%s_s%d" % (c["tag"], s)
            rec[tag + "_q_loss"] = np.asarray(outs[0], np.float64)
            rec[tag + "_q"] = np.asarray(outs[1], np.float64)
            if family == "sqn":
                rec[tag + "_q2"] = np.asarray(outs[2], np.float64)
            _grads(g, names, rec, tag)
            _state(g, names, rec, tag)
        meta["cases"].append(entry)
    _save(family, rec, meta)


def gen_dqn():
    cases = [dict(tag="c0", obs_dim=8, n_actions=4, hidden=[400, 300], batch=128, gamma=0.99, lr=1e-3, polyak=0.995, steps=3, seed=31),
             dict(tag="c1", obs_dim=13, n_actions=3, hidden=[40, 24], batch=19, gamma=0.9, lr=1e-2, polyak=0.9, steps=2, seed=32),
             # config 5's learner shape (algos/dqn with 84x84x4 pixel observations, batch 512)
             dict(tag="c5", obs_dim=84 * 84 * 4, n_actions=4, hidden=[400, 300], batch=512, gamma=0.99, lr=1e-3, polyak=0.995,
                  steps=2, seed=35, pixels=True)]
    _gen_discrete("dqn", os.path.join(REF, "algos", "dqn"), cases, {})


def gen_sqn():
    cases = [dict(tag="c0", obs_dim=8, n_actions=4, hidden=[400, 300], batch=128, gamma=0.99, lr=1e-3, polyak=0.995, steps=3, seed=41),
             dict(tag="c1", obs_dim=13, n_actions=3, hidden=[40, 24], batch=19, gamma=0.9, lr=1e-2, polyak=0.9, steps=2, seed=42),
             dict(tag="c5", obs_dim=84 * 84 * 4, n_actions=4, hidden=[400, 300], batch=512, gamma=0.99, lr=1e-3, polyak=0.995,
                  steps=2, seed=45, pixels=True)]
    _gen_discrete("sqn", os.path.join(REF, "algos", "sqn"), cases, {"alpha": 0.1})


FAMILIES = {"sac1": gen_sac1, "sacv": gen_sacv, "dqn": gen_dqn, "sqn": gen_sqn}


def main(which=None):
    if not os.path.isdir(REF):
        raise SystemExit("reference checkout not present; fixtures can only be regenerated in the build container")
    os.makedirs(OUT, exist_ok=True)
    _install()
    for fam in (which or list(FAMILIES)):
        FAMILIES[fam]()
        print("wrote", fam + "_math.{npz,json}")


if __name__ == "__main__":
    main(sys.argv[1:] or None)
