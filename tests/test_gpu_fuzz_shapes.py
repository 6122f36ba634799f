"""Seeded shape fuzz of the three learners against their float64 oracles: shapes drawn at random inside what the C-ABI accepts —
both the direct-operand envelope and the generic kernels, ragged batches (padding rows), K ranges and column counts that are not
multiples of any tile — first update: losses 1e-5 relative (the north star's tolerance), per-row outputs, gradients.
DDRL_FUZZ_N=<n> draws n shapes per learner instead of the default handful; DDRL_FUZZ_SEED moves the stream."""
import os

import numpy as np
import pytest
import torch

from oracle import dqn_oracle as do
from oracle import sac1_oracle as so

pytestmark = pytest.mark.gpu
N = int(os.environ.get("DDRL_FUZZ_N", "6"))
SEED = int(os.environ.get("DDRL_FUZZ_SEED", "20261003"))


def _rel(a, b):
    return abs(float(a) - float(b)) / max(abs(float(b)), 1e-30)


def _params_close(learner, o64, lr, g, g64, g32):
    """The first Adam step is lr * g / (|g| + eps'): +-lr whatever the size of g, so an element whose gradient is not RELATIVELY
    accurate (a near-zero element next to large ones; the elements behind a flipped relu, see _grads_close) may move the other way.
    Parameters are compared where the gradient is accurate to 1e-3 of itself — which must be nearly everywhere, or as widely as plain
    torch float32 manages on this draw."""
    from distributed_drl_amd import _lib
    ok = np.abs(g - g64) <= 1e-3 * np.abs(g64)
    ok32 = np.abs(g32 - g64) <= 1e-3 * np.abs(g64)
    assert ok.mean() >= min(0.97, 0.95 * ok32.mean()), (ok.mean(), ok32.mean())
    for which, name in ((_lib.SAC1_MAIN, "main"), (_lib.SAC1_TARGET, "target")):
        d = np.abs(learner.export(which).cpu().numpy() - o64.flat(name))
        assert d[ok].max() <= 2e-2 * lr and d.max() <= 2.02 * lr, (name, d[ok].max(), d.max())


def _grads_close(g, g64, g32):
    """2e-4 of the largest gradient element — unless float32 itself does not carry that for this shape: (i) the same math in
    plain torch float32 is off by as much (ill-conditioned draws: a squashed action next to +-1), or (ii) one relu pre-activation
    within rounding of zero has the other sign than in float64 (then one unit's row / column of one kernel differs by one batch
    row's contribution: a vanishing fraction of the elements, each still within 5e-3)."""
    gmax = np.abs(g64).max()
    err, err32 = np.abs(g - g64), np.abs(g32 - g64).max()
    if err.max() <= 2e-4 * gmax or err.max() <= 3.0 * err32:
        return
    assert (err > 2e-4 * gmax).mean() <= 1e-2 and err.max() <= 5e-3 * gmax, (err.max() / gmax, err32 / gmax, (err > 2e-4 * gmax).mean())


def _flip_variants(params, b, cfg, lr, nets, backup, max_cands=4, with_unflipped=False):
    """A relu pre-activation of a differentiated network that is ZERO within float32 rounding has no defined sign in float32: an
    implementation that sums its terms in another order than torch may mask that unit for that batch row where the float64 oracle does
    not (or the other way round).  The forward value moves by ~0, but the gradient moves by that row's whole contribution — a rank-one
    change of every weight gradient below it, which at wide observations is a sizeable fraction of ALL elements and at small batches a
    sizeable fraction of the largest element (what the plain bars of _grads_close / _params_close do not allow for).  This returns, for
    every non-empty subset of the (at most max_cands) entries whose |z| <= 2^-19 sum|terms|, the float64 gradient and first-step
    parameters / targets of the update with those masks inverted — Double-DQN (algos/dqn/actor_learner.py:40-74: nets = ["main/q1"])
    or soft-Q (algos/sqn/actor_learner.py:19-78: the twin networks ["main/q1", "main/q2"], each regressed on the same detached
    `backup`): a float32 result is right if it matches ONE of them."""
    import itertools
    x = b["obs1"].astype(np.float64)
    a = b["acts"].astype(np.int64)
    B = x.shape[0]
    backup = np.asarray(backup, np.float64)
    per_net, cands = [], []
    for ni, pre in enumerate(nets):
        f = lambda k: np.asarray(params[pre + "/" + k], np.float64)
        W1, b1, W2, b2, W3, b3 = f("dense/kernel"), f("dense/bias"), f("dense_1/kernel"), f("dense_1/bias"), f("dense_2/kernel"), f("dense_2/bias")
        z1 = x @ W1 + b1
        h1 = np.maximum(z1, 0)
        z2 = h1 @ W2 + b2
        q = np.maximum(z2, 0) @ W3 + b3
        dq = np.zeros_like(q)
        dq[np.arange(B), a] = (q[np.arange(B), a] - backup) / B
        s1 = np.abs(x) @ np.abs(W1) + np.abs(b1)
        s2 = np.abs(h1) @ np.abs(W2) + np.abs(b2)
        cands += [(ni, 1, i, j) for i, j in zip(*np.where(np.abs(z1) <= 2.0 ** -19 * s1))] + [(ni, 2, i, j) for i, j in zip(*np.where(np.abs(z2) <= 2.0 ** -19 * s2))]
        per_net.append((W1, b1, W2, b2, W3, b3, z1, z2, dq))
    if with_unflipped:     # (tests/test_oracle_math_fixtures.py: the helper's own math against the oracles, no inversion)
        cands = cands[:max_cands]
    elif not cands or len(cands) > max_cands:
        return []
    out = []
    for k in range(0 if with_unflipped else 1, len(cands) + 1):
        for sub in itertools.combinations(cands, k):
            gs, ps = [], []
            for ni, (W1, b1, W2, b2, W3, b3, z1, z2, dq) in enumerate(per_net):
                m1, m2 = z1 > 0, z2 > 0
                for nj, layer, i, j in sub:
                    if nj == ni:
                        (m1 if layer == 1 else m2)[i, j] ^= True
                hh1 = z1 * m1
                hh2 = (hh1 @ W2 + b2) * m2
                dz2 = (dq @ W3.T) * m2
                dz1 = (dz2 @ W2.T) * m1
                gs += [(x.T @ dz1).ravel(), dz1.sum(0), (hh1.T @ dz2).ravel(), dz2.sum(0), (hh2.T @ dq).ravel(), dq.sum(0)]
                ps += [v.ravel() for v in (W1, b1, W2, b2, W3, b3)]
            g, p0 = np.concatenate(gs), np.concatenate(ps)
            main = p0 - lr * g / (np.abs(g) + cfg.adam_eps / np.sqrt(1.0 - cfg.beta2))
            out.append((sub, g, main, cfg.polyak * p0 + (1.0 - cfg.polyak) * main))
    return out


def _sac_shapes(n):
    rs = np.random.RandomState(SEED)
    out = []
    for i in range(n):
        act = int(rs.randint(1, 5)) if i % 3 else int(rs.randint(1, 9))          # > 4 action dims: generic kernels
        obs = int(rs.randint(1, 13 - min(act, 4))) if i % 4 else int(rs.randint(1, 41 - act))
        step = 4 if i % 5 else 1                                                 # hidden % 4 != 0: generic kernels
        h1 = int(rs.randint(1, 512 // step + 1)) * step
        h2 = int(rs.randint(1, 512 // step + 1)) * step
        batch = int(rs.randint(1, 301))
        out.append((obs, act, (h1, h2), batch))
    return out


def _dqn_shapes(n):
    rs = np.random.RandomState(SEED + 1)
    out = []
    for i in range(n):
        acts = 4 if i % 2 == 0 else int(rs.randint(2, 10))
        obs = int(rs.randint(1, 65)) if i % 3 else int(rs.randint(1024, 3000))   # >= 1024: the LDS-DMA layer-1 tiles
        h1 = int(rs.randint(1, 129)) * 4 if i % 4 else int(rs.randint(2, 500))
        h2 = int(rs.randint(1, 129)) * 4 if i % 4 else int(rs.randint(2, 500))
        batch = int(rs.randint(1, 200))
        out.append((obs, acts, (h1, h2), batch, "sqn" if i % 3 == 1 else "ddqn"))
    return out


@pytest.fixture(scope="module")
def ddrl():
    import distributed_drl_amd as d
    d._lib.require_gpu()
    return d


@pytest.mark.parametrize("obs,act,hid,batch", _sac_shapes(N))
def test_sac1_random_shape_first_update(ddrl, obs, act, hid, batch):
    from distributed_drl_amd import _lib
    from distributed_drl_amd.agent import HyperParameters, Learner
    opt = HyperParameters()
    opt.obs_dim, opt.act_dim, opt.hidden_sizes, opt.batch_size, opt.seed = obs, act, hid, batch, 7
    learner = Learner(opt)
    cfg = so.Config(obs_dim=obs, act_dim=act, hidden1=hid[0], hidden2=hid[1], batch=batch, alpha=opt.alpha, gamma=opt.gamma, lr=opt.lr,
                    polyak=opt.polyak)
    params = so.init_params(cfg, 7)
    rs = np.random.RandomState(11)
    for k in params:
        if k.endswith("bias"):
            params[k] = rs.uniform(-0.05, 0.05, params[k].shape).astype(np.float32)
    learner.set_weights(list(params.keys()), list(params.values()))
    o64, o32 = so.Sac1Oracle(cfg, params, torch.float64), so.Sac1Oracle(cfg, params, torch.float32)
    b, eps = so.synthetic_batch(cfg, seed=90)
    w = o64.step(b, *eps)
    w32 = o32.step(b, *eps)
    losses, (q1, q2, lp) = learner.train(b, eps=eps, return_outputs=True)
    for i, k in enumerate(("pi_loss", "q1_loss", "q2_loss")):
        # a loss that is a mean of signed terms can sit near zero: the tolerance is relative to the terms' size then; and a draw with
        # a squashed action next to +-1 (log(1 - pi^2 + 1e-6), core.py:60) is not carried to 1e-5 by float32 at all — there the
        # yardstick is the same formula in plain torch float32 (what the reference's own float32 graph would deliver)
        scale = max(abs(float(w[k])), 1e-2 * float(torch.as_tensor(w["q1"]).abs().mean()))
        tol = max(1e-5 * scale, 2.0 * abs(float(w32[k]) - float(w[k])))
        assert abs(losses[i].item() - float(w[k])) <= tol, (k, losses[i].item(), float(w[k]), float(w32[k]))
    np.testing.assert_allclose(q1.cpu().numpy(), w["q1"].numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(q2.cpu().numpy(), w["q2"].numpy(), rtol=1e-4, atol=1e-5)
    lp64, lp32 = w["logp_pi"].numpy(), w32["logp_pi"].numpy().astype(np.float64)
    assert (np.abs(lp.cpu().numpy() - lp64) <= 2e-5 + 1e-4 * np.abs(lp64) + 3.0 * np.abs(lp32 - lp64)).all()   # row-wise float32 yardstick, as above
    g = learner.export(_lib.SAC1_GRAD).cpu().numpy()
    _grads_close(g, o64.flat("grads"), o32.flat("grads"))
    _params_close(learner, o64, cfg.lr, g, o64.flat("grads"), o32.flat("grads"))
    assert np.isfinite(learner.export(_lib.SAC1_ADAM_V).cpu().numpy()).all()


@pytest.mark.parametrize("obs,acts,hid,batch,variant", _dqn_shapes(N))
def test_dqn_random_shape_first_update(ddrl, obs, acts, hid, batch, variant):
    from distributed_drl_amd import _lib, dqn

    class Opt:
        obs_dim, act_dim, hidden_size, gamma, lr, polyak, batch_size, seed, alpha = obs, acts, list(hid), 0.99, 1e-3, 0.995, batch, 2, 0.1
    sqn = variant == "sqn"
    learner = (dqn.LearnerSQN if sqn else dqn.Learner)(Opt, "learner")
    cfg = do.Config(obs_dim=obs, n_actions=acts, hidden1=hid[0], hidden2=hid[1], batch=batch)
    params = (do.sqn_init_params if sqn else do.init_params)(cfg, 2)
    rs = np.random.RandomState(3)
    for k in params:
        if k.endswith("bias"):
            params[k] = rs.uniform(-0.1, 0.1, params[k].shape).astype(np.float32)
    learner.set_weights(list(params.keys()), list(params.values()))
    o64, o32 = ((do.SqnOracle(cfg, params, 0.1, dt) if sqn else do.DqnOracle(cfg, params, dt)) for dt in (torch.float64, torch.float32))
    b = do.synthetic_batch(cfg, 10)
    w = o64.step(b)
    o32.step(b)
    loss, q = learner.train(b, 0, return_outputs=True)
    assert _rel(loss.item(), w["q_loss"]) <= 1e-5, (loss.item(), float(w["q_loss"]))
    np.testing.assert_allclose(q.cpu().numpy(), w["q"].numpy(), rtol=1e-4, atol=1e-4)
    g = learner.export(_lib.SAC1_GRAD).cpu().numpy()
    try:
        _grads_close(g, o64.flat("grads"), o32.flat("grads"))
        _params_close(learner, o64, cfg.lr, g, o64.flat("grads"), o32.flat("grads"))
    except AssertionError:
        # a relu pre-activation at zero within float32 rounding?  Then the gradient must equal the float64 one under SOME sign
        # assignment of those entries (found by seeds 21 / 23: obs 2560 hidden (68, 340) batch 195 and obs 2398 hidden (280, 240)
        # batch 192 — round 5's library gives the same gradients there)
        # (seed 31: soft-Q, obs 44 hidden (73, 176): one layer-1 unit of one batch row — 44 + 1 elements, 2 % of the largest one)
        main, targ = learner.export(_lib.SAC1_MAIN).cpu().numpy(), learner.export(_lib.SAC1_TARGET).cpu().numpy()
        for sub, g_alt, main_alt, targ_alt in _flip_variants(params, b, cfg, cfg.lr, ["main/q1", "main/q2"] if sqn else ["main/q1"], w["q_backup"].numpy()):
            okg = np.abs(g - g_alt) <= 1e-3 * np.abs(g_alt)
            if np.abs(g - g_alt).max() <= 2e-4 * np.abs(g_alt).max() and okg.mean() >= 0.97 and \
                    np.abs(main - main_alt)[okg].max() <= 2e-2 * cfg.lr and np.abs(targ - targ_alt)[okg].max() <= 2e-2 * cfg.lr:
                break
        else:
            raise
