"""ORACLE — TEST INFRASTRUCTURE ONLY.

CPU restatement (torch, float32 or float64, autograd) of the reference's SAC1 learner update and
policy forward.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this; the product path never does.

Follows:
  network / log-prob / squashing     algos/sac1/core.py:15-18,30-38,45-87,91-121
  losses, optimizers, polyak         algos/sac1/actor_learner.py:26-105
  set_weights -> target_init         algos/sac1/actor_learner.py:125-127
  Actor.get_action                   algos/sac1/actor_learner.py:195-197
  values                             algos/sac1/hyperparams.py:60,67,78,79,82

PARITY: composition PINNED, primitives from memory.  The arithmetic lives in TensorFlow 1.x (tf.layers.dense,
tf.train.AdamOptimizer, tf.random_normal), a third-party dependency that is absent from /root/reference and from this
image (version unpinned; API usage implies 1.12-1.15); the reference holds no tests or golden vectors for it.  What pins
this file: tests/golden/sac1_math.* are produced by EXECUTING the reference's own algos/sac1/actor_learner.py + core.py on
oracle/tf_shim.py (a lazy TF1-graph stand-in over torch float64; oracle/gen_golden_math.py) — losses, q1/q2/logp_pi per row,
every per-variable gradient, parameters / targets / Adam slots after three sequential train() calls from main != target,
Actor.get_action, the variable names and order of get_weights() — and tests/test_oracle_math_fixtures.py holds this oracle to
them at 1e-10 in float64.  So which tensor feeds which loss, the constants, the clip, the squash / scale order, which
variables each optimizer steps and the polyak pairing are the reference's code, executed.  What stays from memory (it is the
shim's definition too, see its header):
  * tf.layers.dense: y = x @ kernel[in,out] + bias; default init glorot-uniform / zeros; variable naming dense, dense_1, ...
  * tf.train.AdamOptimizer (ApplyAdam): alpha_t = lr*sqrt(1-b2^t)/(1-b1^t);
    m += (g-m)*(1-b1); v += (g*g-v)*(1-b2); var -= m*alpha_t/(sqrt(v)+eps);  b^t kept as
    running products in the parameter dtype; one optimizer (own t) per minimize() call.
  * tf.random_normal is replaced by EXPLICIT noise inputs eps_x, eps_x2, eps_t (the graph's 4th draw is never fetched).
  * execution order of one sess.run: all forward values and both gradients from pre-update parameters
    -> Adam(pi) -> Adam(q1,q2) -> polyak with the post-update main.
Self-consistency is checked in tests/test_oracle_sac1.py (closed forms, finite differences in
float64, float32-vs-float64 agreement).
"""
import math
from collections import OrderedDict

import numpy as np
import torch

EPS = 1e-8
LOG_STD_MAX = 2
LOG_STD_MIN = -20


class Config:
    """Defaults = algos/sac1/hyperparams.py at LunarLanderContinuous-v2 dimensions."""

    def __init__(self, obs_dim=8, act_dim=2, hidden1=400, hidden2=300, batch=256, alpha=0.1, gamma=0.997,
                 lr=5e-5, polyak=0.995, beta1=0.9, beta2=0.999, adam_eps=1e-8, act_scale=1.0):
        self.__dict__.update(locals())
        del self.__dict__["self"]


def param_specs(cfg):
    """(name, shape) in TF variable-creation order: main/pi/*, main/q1/*, main/q2/*.
    Names as TF1 assigns them inside each variable scope (dense, dense_1, ...)."""
    o, a, h1, h2 = cfg.obs_dim, cfg.act_dim, cfg.hidden1, cfg.hidden2
    specs = [("main/pi/dense/kernel", (o, h1)), ("main/pi/dense/bias", (h1,)),
             ("main/pi/dense_1/kernel", (h1, h2)), ("main/pi/dense_1/bias", (h2,)),
             ("main/pi/dense_2/kernel", (h2, a)), ("main/pi/dense_2/bias", (a,)),      # mu head
             ("main/pi/dense_3/kernel", (h2, a)), ("main/pi/dense_3/bias", (a,))]      # log_std head
    for q in ("q1", "q2"):
        specs += [("main/%s/dense/kernel" % q, (o + a, h1)), ("main/%s/dense/bias" % q, (h1,)),
                  ("main/%s/dense_1/kernel" % q, (h1, h2)), ("main/%s/dense_1/bias" % q, (h2,)),
                  ("main/%s/dense_2/kernel" % q, (h2, 1)), ("main/%s/dense_2/bias" % q, (1,))]
    return specs


def param_counts(cfg):
    n_pi = sum(int(np.prod(s)) for n, s in param_specs(cfg) if "/pi/" in n)
    n_q = sum(int(np.prod(s)) for n, s in param_specs(cfg) if "/q1/" in n)
    return n_pi, n_q


def init_params(cfg, seed=0):
    """glorot-uniform kernels, zero biases (tf.layers.dense defaults), float32 NumPy, flat order."""
    rs = np.random.RandomState(seed)
    out = OrderedDict()
    for name, shape in param_specs(cfg):
        if name.endswith("kernel"):
            lim = math.sqrt(6.0 / (shape[0] + shape[1]))
            out[name] = rs.uniform(-lim, lim, size=shape).astype(np.float32)
        else:
            out[name] = np.zeros(shape, np.float32)
    return out


def flatten(params):
    return np.concatenate([np.asarray(v, np.float32).reshape(-1) for v in params.values()])


def unflatten(cfg, flat):
    out, off = OrderedDict(), 0
    flat = np.asarray(flat)
    for name, shape in param_specs(cfg):
        n = int(np.prod(shape))
        out[name] = flat[off:off + n].reshape(shape).copy()
        off += n
    assert off == flat.size
    return out


# ------------------------------------------------------------------------------------------
# network math (core.py)
# ------------------------------------------------------------------------------------------
def _dense(x, p, prefix):
    return x @ p[prefix + "/kernel"] + p[prefix + "/bias"]


def _clip_but_pass_gradient(x, l=-1.0, u=1.0):  # core.py:35-38
    clip_up = (x > u).to(x.dtype)
    clip_low = (x < l).to(x.dtype)
    return x + ((u - x) * clip_up + (l - x) * clip_low).detach()


def policy(p, scope, x, eps, cfg, stable=False):
    """mlp_gaussian_policy + apply_squashing_func + action scaling (core.py:49-87,95-106).
    Returns (mu, pi, logp_pi) with mu/pi already scaled by act_scale.
    stable=True evaluates (pi - mu)/(std + EPS) as eps*std/(std + EPS) (identical algebra, since
    pi = mu + eps*std; free of the float32 cancellation of the literal form — the HIP kernel's form)."""
    h = torch.relu(_dense(x, p, scope + "/pi/dense"))
    h = torch.relu(_dense(h, p, scope + "/pi/dense_1"))
    mu = _dense(h, p, scope + "/pi/dense_2")
    log_std = torch.tanh(_dense(h, p, scope + "/pi/dense_3"))
    log_std = LOG_STD_MIN + 0.5 * (LOG_STD_MAX - LOG_STD_MIN) * (log_std + 1)
    std = torch.exp(log_std)
    pi = mu + eps * std
    z = (eps * std) / (std + EPS) if stable else (pi - mu) / (std + EPS)
    pre_sum = -0.5 * (z ** 2 + 2 * log_std + np.log(2 * np.pi))
    logp_pi = pre_sum.sum(dim=1)
    mu = torch.tanh(mu)
    pi = torch.tanh(pi)
    logp_pi = logp_pi - torch.log(_clip_but_pass_gradient(1 - pi ** 2, l=0.0, u=1.0) + 1e-6).sum(dim=1)
    return mu * cfg.act_scale, pi * cfg.act_scale, logp_pi


def qf(p, scope, x, a):
    """vf_mlp(tf.concat([x, a], -1)) (core.py:109-119)."""
    h = torch.cat([x, a], dim=-1)
    h = torch.relu(_dense(h, p, scope + "/dense"))
    h = torch.relu(_dense(h, p, scope + "/dense_1"))
    return _dense(h, p, scope + "/dense_2").squeeze(1)


class Sac1Oracle:
    """Learner (actor_learner.py:19-148) with explicit noise; dtype float32 or float64."""

    def __init__(self, cfg, params, dtype=torch.float32, stable=False):
        self.cfg, self.dtype, self.stable = cfg, dtype, stable
        self.names = [n for n, _ in param_specs(cfg)]
        self.main = OrderedDict((n, torch.tensor(np.asarray(params[n]), dtype=dtype).clone()) for n in self.names)
        self.set_weights(self.names, [self.main[n] for n in self.names])
        zeros = lambda: OrderedDict((n, torch.zeros_like(v)) for n, v in self.main.items())
        self.m, self.v = zeros(), zeros()
        # running beta powers, one pair per optimizer (pi, value), kept in the parameter dtype
        one = torch.tensor(1.0, dtype=dtype)
        self.b1p = {"pi": one * cfg.beta1, "q": one * cfg.beta1}
        self.b2p = {"pi": one * cfg.beta2, "q": one * cfg.beta2}
        self.grads = None

    # actor_learner.py:125-127
    def set_weights(self, names, values):
        for n, v in zip(names, values):
            self.main[n] = torch.as_tensor(np.asarray(v) if not torch.is_tensor(v) else v).to(self.dtype).clone()
        self.target = OrderedDict((n.replace("main/", "target/", 1), v.clone()) for n, v in self.main.items())

    def get_weights(self):
        return list(self.main.keys()), [v.numpy().copy() for v in self.main.values()]

    def _t(self, x):
        return torch.as_tensor(np.asarray(x)).to(self.dtype)

    def forward_losses(self, batch, eps_x, eps_x2, eps_t, main=None):
        cfg = self.cfg
        p = self.main if main is None else main
        x, x2, a = self._t(batch["obs1"]), self._t(batch["obs2"]), self._t(batch["acts"])
        r, d = self._t(batch["rews"]), self._t(batch["done"])
        eps_x, eps_x2, eps_t = self._t(eps_x), self._t(eps_x2), self._t(eps_t)
        mu, pi, logp_pi = policy(p, "main", x, eps_x, cfg, self.stable)
        _, _, logp_pi2 = policy(p, "main", x2, eps_x2, cfg, self.stable)
        q1 = qf(p, "main/q1", x, a)
        q2 = qf(p, "main/q2", x, a)
        q1_pi = qf(p, "main/q1", x, pi)
        # target network: a full actor-critic copy evaluated at x2 (actor_learner.py:36-38)
        _, pi_t, _ = policy(self.target, "target", x2, eps_t, cfg, self.stable)
        q1_pi_t = qf(self.target, "target/q1", x2, pi_t)
        q2_pi_t = qf(self.target, "target/q2", x2, pi_t)
        min_q_pi = torch.minimum(q1_pi_t, q2_pi_t)
        v_backup = (min_q_pi - cfg.alpha * logp_pi2).detach()
        q_backup = r + cfg.gamma * (1 - d) * v_backup
        pi_loss = (cfg.alpha * logp_pi - q1_pi).mean()
        q1_loss = 0.5 * ((q_backup - q1) ** 2).mean()
        q2_loss = 0.5 * ((q_backup - q2) ** 2).mean()
        return dict(pi_loss=pi_loss, q1_loss=q1_loss, q2_loss=q2_loss, q1=q1, q2=q2, logp_pi=logp_pi,
                    pi=pi, mu=mu, logp_pi2=logp_pi2, q1_pi=q1_pi, q_backup=q_backup, pi_targ=pi_t)

    def compute_grads(self, batch, eps_x, eps_x2, eps_t):
        leaves = OrderedDict((n, v.clone().requires_grad_(True)) for n, v in self.main.items())
        out = self.forward_losses(batch, eps_x, eps_x2, eps_t, main=leaves)
        pi_names = [n for n in self.names if "/pi/" in n]
        q_names = [n for n in self.names if "/q1/" in n or "/q2/" in n]
        g_pi = torch.autograd.grad(out["pi_loss"], [leaves[n] for n in pi_names], retain_graph=True)
        g_q = torch.autograd.grad(out["q1_loss"] + out["q2_loss"], [leaves[n] for n in q_names])
        self.grads = OrderedDict(list(zip(pi_names, g_pi)) + list(zip(q_names, g_q)))
        return {k: (v.detach() if torch.is_tensor(v) else v) for k, v in out.items()}

    def _adam(self, names, which):
        c = self.cfg
        one = torch.tensor(1.0, dtype=self.dtype)
        lr = torch.tensor(c.lr, dtype=self.dtype)
        alpha_t = lr * torch.sqrt(one - self.b2p[which]) / (one - self.b1p[which])
        b1 = torch.tensor(c.beta1, dtype=self.dtype)
        b2 = torch.tensor(c.beta2, dtype=self.dtype)
        eps = torch.tensor(c.adam_eps, dtype=self.dtype)
        for n in names:
            g = self.grads[n]
            self.m[n] = self.m[n] + (g - self.m[n]) * (one - b1)
            self.v[n] = self.v[n] + (g * g - self.v[n]) * (one - b2)
            self.main[n] = self.main[n] - (self.m[n] * alpha_t) / (torch.sqrt(self.v[n]) + eps)
        self.b1p[which] = self.b1p[which] * b1
        self.b2p[which] = self.b2p[which] * b2

    def apply_grads(self):
        c = self.cfg
        self._adam([n for n in self.names if "/pi/" in n], "pi")
        self._adam([n for n in self.names if "/q1/" in n or "/q2/" in n], "q")
        pk = torch.tensor(c.polyak, dtype=self.dtype)
        pk1 = torch.tensor(1 - c.polyak, dtype=self.dtype)
        for n in self.names:
            tn = n.replace("main/", "target/", 1)
            self.target[tn] = pk * self.target[tn] + pk1 * self.main[n]

    def step(self, batch, eps_x, eps_x2, eps_t):
        """== Learner.train(batch): returns step_ops[0:6] values from pre-update parameters."""
        out = self.compute_grads(batch, eps_x, eps_x2, eps_t)
        self.apply_grads()
        return out

    def flat(self, which="main"):
        d = {"main": self.main, "target": self.target, "m": self.m, "v": self.v, "grads": self.grads}[which]
        return np.concatenate([v.detach().numpy().reshape(-1) for v in d.values()])


def actor_act(cfg, params, obs, eps, deterministic=False, dtype=torch.float32):
    """Actor.get_action batched (actor_learner.py:195-197): pi (or mu) of the main policy."""
    p = OrderedDict((n, torch.as_tensor(np.asarray(v)).to(dtype)) for n, v in params.items() if "/pi/" in n)
    x = torch.as_tensor(np.asarray(obs)).to(dtype).reshape(-1, cfg.obs_dim)
    e = torch.zeros(x.shape[0], cfg.act_dim, dtype=dtype) if eps is None else torch.as_tensor(np.asarray(eps)).to(dtype)
    mu, pi, _ = policy(p, "main", x, e, cfg)
    return (mu if deterministic else pi).numpy()


def synthetic_batch(cfg, seed=1234, n=None):
    """SURVEY §8(d) synthetic transitions + explicit noise."""
    n = cfg.batch if n is None else n
    rs = np.random.RandomState(seed)
    batch = dict(obs1=rs.randn(n, cfg.obs_dim).astype(np.float32), obs2=rs.randn(n, cfg.obs_dim).astype(np.float32),
                 acts=rs.uniform(-1, 1, (n, cfg.act_dim)).astype(np.float32), rews=rs.randn(n).astype(np.float32),
                 done=(rs.rand(n) < 0.01).astype(np.float32))
    rn = np.random.RandomState(seed + 1)
    eps = [rn.randn(n, cfg.act_dim).astype(np.float32) for _ in range(3)]
    return batch, eps
