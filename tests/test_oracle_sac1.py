"""Self-consistency of the SAC1 oracle (parity vs TensorFlow is unpinned — see its header):
closed forms, float32-vs-float64 agreement, finite-difference gradients in float64, the TF1 Adam
formula on a scalar, polyak / target_init semantics, parameter layout counts."""
import math

import numpy as np
import torch

from oracle import sac1_oracle as so


def test_param_layout_counts():
    cfg = so.Config()
    n_pi, n_q = so.param_counts(cfg)
    assert (n_pi, n_q) == (125104, 125001)  # 375 106 in total
    specs = so.param_specs(cfg)
    assert len(specs) == 20 and len([n for n, _ in specs if "/pi/" in n]) == 8
    p = so.init_params(cfg, 0)
    flat = so.flatten(p)
    assert flat.size == n_pi + 2 * n_q
    back = so.unflatten(cfg, flat)
    assert all((back[k] == p[k]).all() for k in p)
    lim = math.sqrt(6.0 / (400 + 300))
    assert abs(p["main/pi/dense_1/kernel"]).max() <= lim and (p["main/q1/dense/bias"] == 0).all()


def test_zero_weight_closed_form():
    """All-zero weights: mu=0, log_std=-20+11=-9 (core.py:73-74); logp closed form."""
    cfg = so.Config(batch=4)
    p = {k: np.zeros_like(v) for k, v in so.init_params(cfg).items()}
    o = so.Sac1Oracle(cfg, p, torch.float64)
    batch, eps = so.synthetic_batch(cfg, n=4)
    out = o.forward_losses(batch, *eps)
    e = torch.tensor(eps[0], dtype=torch.float64)
    std = math.exp(-9.0)
    z = e * std / (std + 1e-8)
    gauss = (-0.5 * (z ** 2 + 2 * (-9.0) + math.log(2 * math.pi))).sum(1)
    a = torch.tanh(e * std)
    want = gauss - torch.log(torch.clamp(1 - a ** 2, 0, 1) + 1e-6).sum(1)
    assert torch.allclose(out["logp_pi"], want, rtol=0, atol=1e-12)
    assert float(out["q1"].abs().max()) == 0.0
    # q_backup = r + gamma*(1-d)*(0 - alpha*logp_pi2); q-loss = 0.5*mean(backup^2)
    assert abs(float(out["q1_loss"]) - 0.5 * float((out["q_backup"] ** 2).mean())) < 1e-12


def test_fp32_vs_fp64_losses_agree():
    """The reference's literal float32 formula (pi - mu)/(std + EPS) cancels catastrophically
    (pi = mu + eps*std with std ~ 1e-4): per-row logp is noisy at ~5e-3 and the q-losses at ~1e-5
    relative — the float32 noise floor of the reference formulation.  The algebraically identical
    form eps*std/(std + EPS) (the HIP kernel's) removes it."""
    cfg = so.Config()
    p = so.init_params(cfg, 0)
    batch, eps = so.synthetic_batch(cfg)
    o32, o64 = so.Sac1Oracle(cfg, p, torch.float32), so.Sac1Oracle(cfg, p, torch.float64)
    s32 = so.Sac1Oracle(cfg, p, torch.float32, stable=True)
    s64 = so.Sac1Oracle(cfg, p, torch.float64, stable=True)
    a, b, c, d = o32.step(batch, *eps), o64.step(batch, *eps), s32.step(batch, *eps), s64.step(batch, *eps)
    rel = lambda x, y: abs(float(x) - float(y)) / abs(float(y))
    for k in ("pi_loss", "q1_loss", "q2_loss"):
        assert rel(d[k], b[k]) <= 1e-12, k            # same algebra in float64
        assert rel(c[k], b[k]) <= 2e-6, k             # stable float32 form: at rounding level
        assert rel(a[k], b[k]) <= 5e-5, k             # literal float32 form: its own noise floor
    assert float((a["logp_pi"].double() - b["logp_pi"]).abs().max()) > 1e-4   # the cancellation is real
    assert float((c["logp_pi"].double() - b["logp_pi"]).abs().max()) < 2e-5
    g32, g64 = s32.flat("grads"), o64.flat("grads")
    assert np.abs(g32 - g64).max() <= 1e-5 * np.abs(g64).max()


def test_gradients_by_finite_differences_fp64():
    cfg = so.Config(hidden1=12, hidden2=9, batch=6)
    p = so.init_params(cfg, 3)
    for k in p:  # non-zero biases so every path is exercised
        if k.endswith("bias"):
            p[k] = np.random.RandomState(1).uniform(-0.1, 0.1, p[k].shape).astype(np.float32)
    batch, eps = so.synthetic_batch(cfg, n=6)
    o = so.Sac1Oracle(cfg, p, torch.float64, stable=True)  # the literal form's cancellation noise (~1e-12) / h would swamp the difference quotient
    o.compute_grads(batch, *eps)
    rs = np.random.RandomState(0)
    h = 1e-6
    for name in o.names:
        loss_key = ("pi_loss",) if "/pi/" in name else ("q1_loss", "q2_loss")
        for _ in range(3):
            idx = tuple(rs.randint(0, s) for s in o.main[name].shape)
            base = o.main[name][idx].item()
            vals = []
            for sgn in (+1, -1):
                o.main[name][idx] = base + sgn * h
                out = o.forward_losses(batch, *eps)
                vals.append(sum(float(out[k]) for k in loss_key))
            o.main[name][idx] = base
            fd = (vals[0] - vals[1]) / (2 * h)
            an = float(o.grads[name][idx])
            assert abs(fd - an) <= 1e-6 + 1e-5 * abs(an), (name, idx, fd, an)


def test_pi_gradient_ignores_q_params_and_value_gradient_ignores_pi():
    cfg = so.Config(hidden1=8, hidden2=8, batch=4)
    o = so.Sac1Oracle(cfg, so.init_params(cfg, 1), torch.float64)
    batch, eps = so.synthetic_batch(cfg, n=4)
    o.compute_grads(batch, *eps)
    assert set(o.grads) == set(o.names)


def test_tf1_adam_formula_scalar():
    """One variable, known gradients: m += (g-m)(1-b1); v += (g^2-v)(1-b2);
    var -= lr*sqrt(1-b2^t)/(1-b1^t) * m/(sqrt(v)+eps) — epsilon OUTSIDE the sqrt."""
    cfg = so.Config(hidden1=2, hidden2=2, batch=2, lr=0.1)
    o = so.Sac1Oracle(cfg, so.init_params(cfg, 1), torch.float64)
    name = o.names[0]
    w0 = o.main[name].clone()
    m = v = 0.0
    w = w0.clone()
    for t, gval in enumerate((0.5, -0.25, 0.125), start=1):
        o.grads = {n: torch.full_like(p, gval) for n, p in o.main.items()}
        o.apply_grads()
        m = 0.9 * m + 0.1 * gval
        v = 0.999 * v + 0.001 * gval * gval
        lr_t = 0.1 * math.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t)
        w = w - lr_t * m / (math.sqrt(v) + 1e-8)
        assert torch.allclose(o.main[name], w, rtol=1e-12, atol=1e-14)


def test_polyak_and_target_init():
    cfg = so.Config(hidden1=4, hidden2=4, batch=2)
    o = so.Sac1Oracle(cfg, so.init_params(cfg, 2), torch.float64)
    n = o.names[2]
    assert torch.equal(o.target[n.replace("main/", "target/")], o.main[n])  # target_init at set_weights
    t0 = o.target[n.replace("main/", "target/")].clone()
    batch, eps = so.synthetic_batch(cfg, n=2)
    o.step(batch, *eps)
    want = 0.995 * t0 + (1 - 0.995) * o.main[n]  # polyak uses the POST-update main
    assert torch.allclose(o.target[n.replace("main/", "target/")], want, rtol=0, atol=1e-15)
    o.set_weights([n], [o.main[n].numpy()])
    assert torch.equal(o.target[n.replace("main/", "target/")], o.main[n])


def test_actor_act_matches_learner_policy():
    cfg = so.Config()
    p = so.init_params(cfg, 0)
    batch, eps = so.synthetic_batch(cfg)
    o = so.Sac1Oracle(cfg, p, torch.float32)
    out = o.forward_losses(batch, *eps)
    act = so.actor_act(cfg, p, batch["obs1"], eps[0])
    np.testing.assert_array_equal(act, out["pi"].numpy())
    det = so.actor_act(cfg, p, batch["obs1"], None, deterministic=True)
    np.testing.assert_array_equal(det, out["mu"].numpy())
    assert np.abs(act).max() <= 1.0


def test_sacv_oracle_gradients_match_finite_differences():
    """example/model.py:33-52 restated: autograd gradients of pi_loss (w.r.t. main/pi) and of
    q1_loss + q2_loss + v_loss (w.r.t. main/q1, q2, v) agree with central differences in float64;
    the stop_gradient targets carry no gradient (pi_loss does not move q / v variables and v.v.)."""
    from oracle import sacv_oracle as sv
    cfg = so.Config(obs_dim=4, act_dim=2, hidden1=12, hidden2=8, batch=6, alpha=0.2, gamma=0.99, lr=1e-3)
    params = sv.init_params(cfg, 3)
    rs = np.random.RandomState(0)
    for k in params:
        if k.endswith("bias"):
            params[k] = rs.uniform(-0.3, 0.3, params[k].shape).astype(np.float32)
    batch, eps = so.synthetic_batch(cfg, seed=2)
    o = sv.SacVOracle(cfg, params, torch.float64, stable=True)
    base_out = o.compute_grads(batch, eps[0])
    frozen = (base_out["q_backup"], base_out["v_backup"])
    names = list(o.names)
    pick = rs.choice(len(names), 10, replace=False)
    for ni in pick:
        n = names[ni]
        g = o.grads[n].numpy().reshape(-1)
        idx = rs.randint(0, g.size)
        base = o.main[n].clone()
        vals = []
        for sgn in (+1, -1):
            t = base.clone().reshape(-1)
            t[idx] += sgn * 1e-6
            o.main[n] = t.reshape(base.shape)
            out = o.forward_losses(batch, eps[0], frozen=frozen)
            vals.append(float(out["pi_loss"]) if "/pi/" in n else float(out["q1_loss"] + out["q2_loss"] + out["v_loss"]))
        o.main[n] = base
        fd = (vals[0] - vals[1]) / 2e-6
        assert abs(fd - g[idx]) <= 1e-6 + 1e-5 * abs(fd), (n, fd, g[idx])
    # the first Adam step moves every variable with a gradient by ~lr; polyak keeps target within (1-polyak)*lr
    before = o.flat("main").copy()
    o.apply_grads()
    assert np.abs(o.flat("main") - before).max() <= 1.01 * cfg.lr
    assert len(o.names) == 26 and sum("/v/" in n for n in o.names) == 6


def test_dqn_oracle_gradients_match_finite_differences():
    """algos/dqn/actor_learner.py:40-62 restated: d q_loss / d main variables (Double-DQN target frozen, as
    tf.stop_gradient) agrees with central differences in float64."""
    from oracle import dqn_oracle as do
    cfg = do.Config(obs_dim=5, n_actions=3, hidden1=10, hidden2=7, batch=9)
    params = do.init_params(cfg, 1)
    rs = np.random.RandomState(0)
    for k in params:
        if k.endswith("bias"):
            params[k] = rs.uniform(-0.3, 0.3, params[k].shape).astype(np.float32)
    batch = do.synthetic_batch(cfg, 4)
    o = do.DqnOracle(cfg, params, torch.float64)
    base = o.forward_loss(batch)
    ref = do.DqnOracle(cfg, params, torch.float64)
    ref.step(batch)
    for n in o.names:
        g = ref.grads[n].numpy().reshape(-1)
        idx = rs.randint(0, g.size)
        keep = o.main[n].clone()
        vals = []
        for sgn in (+1, -1):
            t = keep.clone().reshape(-1)
            t[idx] += sgn * 1e-6
            o.main[n] = t.reshape(keep.shape)
            vals.append(float(o.forward_loss(batch, frozen=base["q_backup"])["q_loss"]))
        o.main[n] = keep
        fd = (vals[0] - vals[1]) / 2e-6
        assert abs(fd - g[idx]) <= 1e-7 + 1e-5 * abs(fd), (n, fd, g[idx])


def test_sqn_oracle_gradients_match_finite_differences():
    """algos/sqn/actor_learner.py:40-56: d q_loss / d (main/q1, main/q2) with v_backup frozen (tf.stop_gradient:
    the softmax term at x2 depends on main/q1) agrees with central differences in float64."""
    from oracle import dqn_oracle as do
    cfg = do.Config(obs_dim=5, n_actions=3, hidden1=10, hidden2=7, batch=9)
    params = do.sqn_init_params(cfg, 1)
    rs = np.random.RandomState(0)
    for k in params:
        if k.endswith("bias"):
            params[k] = rs.uniform(-0.3, 0.3, params[k].shape).astype(np.float32)
    batch = do.synthetic_batch(cfg, 4)
    o = do.SqnOracle(cfg, params, 0.1, torch.float64)
    base = o.forward_loss(batch)
    ref = do.SqnOracle(cfg, params, 0.1, torch.float64)
    ref.step(batch)
    assert len(o.names) == 12
    for n in o.names:
        g = ref.grads[n].numpy().reshape(-1)
        idx = rs.randint(0, g.size)
        keep = o.main[n].clone()
        vals = []
        for sgn in (+1, -1):
            t = keep.clone().reshape(-1)
            t[idx] += sgn * 1e-6
            o.main[n] = t.reshape(keep.shape)
            vals.append(float(o.forward_loss(batch, frozen=base["q_backup"])["q_loss"]))
        o.main[n] = keep
        fd = (vals[0] - vals[1]) / 2e-6
        assert abs(fd - g[idx]) <= 1e-7 + 1e-5 * abs(fd), (n, fd, g[idx])
