"""The float oracles against golden vectors produced by EXECUTING the reference's own learner text
(oracle/gen_golden_math.py: algos/sac1/actor_learner.py + core.py, example/model.py + core.py,
algos/dqn and algos/sqn actor_learner.py + core.py, running on oracle/tf_shim.py in float64).

Bar: the oracle in float64 reproduces losses, per-row outputs, every per-variable gradient, and
parameters / targets / Adam slots after each of the sequential train() calls to 1e-10 relative —
both sides are float64 evaluations of the same formulas, so what this catches is a difference in
COMPOSITION (which tensor feeds which loss, constants, clip, squash/scale order, which variables
an optimizer steps, the polyak pairing, the variable order of get_weights)."""
import json
import os
from collections import OrderedDict

import numpy as np
import pytest
import torch

from oracle import dqn_oracle as dq
from oracle import fixture_inputs as fi
from oracle import sac1_oracle as so
from oracle import sacv_oracle as sv

GOLD = os.path.join(os.path.dirname(__file__), "golden")
RTOL = 1e-10


def load(family):
    with open(os.path.join(GOLD, family + "_math.json")) as f:
        meta = json.load(f)
    return meta, np.load(os.path.join(GOLD, family + "_math.npz"))


def case_params(c):
    names, shapes = c["names"], [tuple(s) for s in c["shapes"]]
    main = fi.make_params(list(zip(names, shapes)), c["seed"], "main")
    targ = fi.make_params(list(zip(names, shapes)), c["seed"], "target")
    if c.get("pixels"):
        main[0] = main[0] * np.float32(1.0 / 64)
        targ[0] = targ[0] * np.float32(1.0 / 64)
    return names, main, targ


def set_targets(o, names, targ):
    for n, t in zip(names, targ):
        o.target[n.replace("main/", "target/", 1)] = torch.tensor(t, dtype=o.dtype)


def check_state(o, z, tag, names, rtol=RTOL):
    for i, n in enumerate(names):
        for what, d in (("grad", o.grads), ("main", o.main), ("m", o.m), ("v", o.v)):
            ok, err, norm = fi.digest_close(d[n].detach().numpy(), z["%s_%s_%d" % (tag, what, i)], rtol, rtol)
            assert ok, (tag, what, n, err, norm)
        ok, err, norm = fi.digest_close(o.target[n.replace("main/", "target/", 1)].numpy(), z["%s_targ_%d" % (tag, i)], rtol, rtol)
        assert ok, (tag, "target", n, err, norm)


def close(a, b, rtol=RTOL):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() <= rtol * max(1.0, np.abs(b).max())


def test_sac1_oracle_equals_reference_text():
    meta, z = load("sac1")
    assert meta["noise_calls"][3].startswith("target pi @ x2, second")
    for c in meta["cases"]:
        cfg = so.Config(obs_dim=c["obs_dim"], act_dim=c["act_dim"], batch=c["batch"], alpha=c["alpha"], gamma=c["gamma"],
                        lr=c["lr"], polyak=c["polyak"], act_scale=c["act_high"])
        names, main, targ = case_params(c)
        # variable names and get_weights order are the reference's (through the shim's TensorFlowVariables)
        assert names == [n for n, _ in so.param_specs(cfg)]
        assert c["actor_names"] == [n for n in names if "/pi/" in n]
        assert c["optimizer_var_lists"] == [[n for n in names if "/pi/" in n], [n for n in names if "/q" in n]]
        assert all(u == [0, 1, 2] for u in c["noise_used"])          # the 4th normal draw is never fetched
        o = so.Sac1Oracle(cfg, OrderedDict(zip(names, main)), torch.float64)
        set_targets(o, names, targ)
        for s in range(c["steps"]):
            batch, noise = fi.sac_batch(c["obs_dim"], c["act_dim"], c["batch"], 100 * c["seed"] + s, c["act_high"])
            out = o.step(batch, *noise[:3])
            tag = "%s_s%d" % (c["tag"], s)
            for k in ("pi_loss", "q1_loss", "q2_loss", "q1", "q2", "logp_pi"):
                assert close(out[k].numpy(), z[tag + "_" + k]), (tag, k)
            assert float(z[tag + "_alpha"]) == c["alpha"]
            check_state(o, z, tag, names)
        # Actor.get_action, stochastic and deterministic, one observation per call in the reference
        p = OrderedDict(zip(names, main))
        obs, eps = z[c["tag"] + "_actor_obs"], z[c["tag"] + "_actor_eps"]
        assert close(so.actor_act(cfg, p, obs, eps, dtype=torch.float64), z[c["tag"] + "_actor_pi"])
        assert close(so.actor_act(cfg, p, obs, None, deterministic=True, dtype=torch.float64), z[c["tag"] + "_actor_mu"])


def test_sac1_stable_form_is_the_same_function():
    """The kernels evaluate eps*std/(std+EPS) for the reference's (pi-mu)/(std+EPS): same values in float64."""
    meta, z = load("sac1")
    c = meta["cases"][0]
    cfg = so.Config(obs_dim=c["obs_dim"], act_dim=c["act_dim"], batch=c["batch"], alpha=c["alpha"], gamma=c["gamma"],
                    lr=c["lr"], polyak=c["polyak"], act_scale=c["act_high"])
    names, main, targ = case_params(c)
    o = so.Sac1Oracle(cfg, OrderedDict(zip(names, main)), torch.float64, stable=True)
    set_targets(o, names, targ)
    batch, noise = fi.sac_batch(c["obs_dim"], c["act_dim"], c["batch"], 100 * c["seed"], c["act_high"])
    out = o.step(batch, *noise[:3])
    for k in ("pi_loss", "q1_loss", "q2_loss", "logp_pi"):
        assert close(out[k].numpy(), z["c0_s0_" + k], 1e-9), k
    check_state(o, z, "c0_s0", names, 1e-7)


def test_sacv_oracle_equals_reference_text():
    meta, z = load("sacv")
    for c in meta["cases"]:
        cfg = so.Config(obs_dim=c["obs_dim"], act_dim=c["act_dim"], hidden1=c["hid"], hidden2=c["hid"], batch=c["batch"],
                        alpha=c["alpha"], gamma=c["gamma"], lr=c["lr"], polyak=c["polyak"], act_scale=c["act_high"])
        names, main, targ = case_params(c)
        assert names == [n for n, _ in sv.param_specs(cfg)]
        assert c["n_random"] == 2 and all(u == [0] for u in c["noise_used"])   # main pi @ x only; the target's is unused
        assert c["optimizer_var_lists"][1] == [n for n in names if "/q" in n or "/v/" in n]
        o = sv.SacVOracle(cfg, OrderedDict(zip(names, main)), torch.float64)
        set_targets(o, names, targ)
        for s in range(c["steps"]):
            batch, noise = fi.sac_batch(c["obs_dim"], c["act_dim"], c["batch"], 100 * c["seed"] + s, c["act_high"])
            out = o.step(batch, noise[0])
            tag = "%s_s%d" % (c["tag"], s)
            for k in ("pi_loss", "q1_loss", "q2_loss", "v_loss", "q1", "q2", "v", "logp_pi"):
                assert close(out[k].numpy(), z[tag + "_" + k]), (tag, k)
            check_state(o, z, tag, names)


@pytest.mark.parametrize("family", ["dqn", "sqn"])
def test_discrete_oracles_equal_reference_text(family):
    meta, z = load(family)
    for c in meta["cases"]:
        cfg = dq.Config(obs_dim=c["obs_dim"], n_actions=c["n_actions"], hidden1=c["hidden"][0], hidden2=c["hidden"][1],
                        batch=c["batch"], gamma=c["gamma"], lr=c["lr"], polyak=c["polyak"])
        names, main, targ = case_params(c)
        specs = dq.param_specs(cfg) if family == "dqn" else dq.sqn_param_specs(cfg)
        assert names == [n for n, _ in specs]
        assert c["optimizer_var_lists"] == [names]
        p = OrderedDict(zip(names, main))
        o = dq.DqnOracle(cfg, p, torch.float64) if family == "dqn" else dq.SqnOracle(cfg, p, 0.1, torch.float64)
        set_targets(o, names, targ)
        for s in range(c["steps"]):
            batch = fi.dqn_batch(c["obs_dim"], c["n_actions"], c["batch"], 100 * c["seed"] + s, c.get("pixels", False))
            out = o.step(batch)
            tag = "%s_s%d" % (c["tag"], s)
            assert close(out["q_loss"].numpy(), z[tag + "_q_loss"]), tag
            assert close(out["q"].numpy(), z[tag + "_q"]), tag
            if family == "sqn":
                assert close(out["q2"].numpy(), z[tag + "_q2"]), tag
            check_state(o, z, tag, names)


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="build container only: regenerates from the reference checkout")
def test_fixtures_regenerate_identically_from_the_reference(tmp_path):
    """The committed vectors are what the reference's text produces today: rerun the generator (small families) into a
    scratch directory and compare every array bit for bit."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r); from oracle import gen_golden_math as g; g.OUT = %r; g.main(['sac1', 'sacv'])"
            % (root, str(tmp_path)))
    subprocess.check_call([sys.executable, "-c", code], stdout=subprocess.DEVNULL)
    for fam in ("sac1", "sacv"):
        a, b = np.load(os.path.join(GOLD, fam + "_math.npz")), np.load(os.path.join(str(tmp_path), fam + "_math.npz"))
        assert sorted(a.files) == sorted(b.files)
        for k in a.files:
            assert np.array_equal(a[k], b[k]), (fam, k)
        with open(os.path.join(GOLD, fam + "_math.json")) as f, open(os.path.join(str(tmp_path), fam + "_math.json")) as g2:
            assert json.load(f) == json.load(g2)


@pytest.mark.parametrize("sqn", [False, True])
def test_relu_sign_variants_of_the_fuzz_test_are_the_oracles_own_math(sqn):
    """tests/test_gpu_fuzz_shapes.py accepts a HIP gradient that equals the float64 one under ANOTHER sign assignment of the relu
    pre-activations that are zero within float32 rounding (_flip_variants: its own NumPy restatement of the Double-DQN / soft-Q
    gradient and first Adam step).  That restatement with NO sign inverted must be the float64 oracle's result, and inverting one mask
    entry must change exactly the gradient elements below that unit."""
    from test_gpu_fuzz_shapes import _flip_variants
    cfg = dq.Config(obs_dim=44, n_actions=2, hidden1=73, hidden2=176, batch=37)
    params = (dq.sqn_init_params if sqn else dq.init_params)(cfg, 2)
    rs = np.random.RandomState(3)
    for k in params:
        if k.endswith("bias"):
            params[k] = rs.uniform(-0.1, 0.1, params[k].shape).astype(np.float32)
    o64 = dq.SqnOracle(cfg, params, 0.1, torch.float64) if sqn else dq.DqnOracle(cfg, params, torch.float64)
    b = dq.synthetic_batch(cfg, 10)
    w = o64.step(b)
    nets = ["main/q1", "main/q2"] if sqn else ["main/q1"]
    # (a threshold of 2^-19 finds no candidate on this draw: hand the helper the entry closest to zero by scaling nothing — the
    # unflipped form is returned first whatever the candidates are)
    out = _flip_variants(params, b, cfg, cfg.lr, nets, w["q_backup"].numpy(), max_cands=0, with_unflipped=True)
    assert len(out) == 1 and out[0][0] == ()
    _, g, main, targ = out[0]
    np.testing.assert_allclose(g, o64.flat("grads"), rtol=0, atol=1e-15)
    np.testing.assert_allclose(main, o64.flat("main"), rtol=0, atol=1e-12)
    np.testing.assert_allclose(targ, o64.flat("target"), rtol=0, atol=1e-12)
