"""ORACLE — TEST INFRASTRUCTURE ONLY.

CPU restatement (torch autograd, float32 / float64) of the SAC-v learner of example/model.py:17-76
(policy + twin Q + V + target V; the algorithm example/dsac.py actually trains) on the network of
example/core.py:45-127.  Only tests/ may import this; the product path never does.

Follows:
  mlp_actor_critic (pi, q1, q2, q1_pi, q2_pi, v)     example/core.py:98-127
  min double-Q, q_backup, v_backup, the four losses   example/model.py:33-45
  Adam(pi) -> Adam(q1, q2, v) -> polyak(all main)     example/model.py:47-64
PARITY: composition PINNED by tests/golden/sacv_math.* — example/model.py + example/core.py executed on oracle/tf_shim.py with
example/dsac.py's own args (oracle/gen_golden_math.py); tests/test_oracle_math_fixtures.py holds this oracle to them at 1e-10
in float64.  The TF primitives (dense, ApplyAdam, variable naming) stay from memory, as in oracle/sac1_oracle.py.
"""
from collections import OrderedDict

import numpy as np
import torch

from .sac1_oracle import Config, Sac1Oracle, _dense, policy, qf, param_specs as _sac1_specs


def param_specs(cfg):
    o, h1, h2 = cfg.obs_dim, cfg.hidden1, cfg.hidden2
    return _sac1_specs(cfg) + [("main/v/dense/kernel", (o, h1)), ("main/v/dense/bias", (h1,)),
                               ("main/v/dense_1/kernel", (h1, h2)), ("main/v/dense_1/bias", (h2,)),
                               ("main/v/dense_2/kernel", (h2, 1)), ("main/v/dense_2/bias", (1,))]


def init_params(cfg, seed=0):
    """glorot-uniform kernels / zero biases in variable-creation order (one RandomState stream, as agent.glorot_init)."""
    rs = np.random.RandomState(seed)
    out = OrderedDict()
    for name, shape in param_specs(cfg):
        if name.endswith("kernel"):
            lim = np.sqrt(6.0 / (shape[0] + shape[1]))
            out[name] = rs.uniform(-lim, lim, size=shape).astype(np.float32)
        else:
            out[name] = np.zeros(shape, np.float32)
    return out


def vf(p, scope, x):
    h = torch.relu(_dense(x, p, scope + "/dense"))
    h = torch.relu(_dense(h, p, scope + "/dense_1"))
    return _dense(h, p, scope + "/dense_2").squeeze(1)


class SacVOracle(Sac1Oracle):
    def __init__(self, cfg, params, dtype=torch.float32, stable=False):
        self.cfg, self.dtype, self.stable = cfg, dtype, stable
        self.names = [n for n, _ in param_specs(cfg)]
        self.main = OrderedDict((n, torch.tensor(np.asarray(params[n]), dtype=dtype).clone()) for n in self.names)
        self.set_weights(self.names, [self.main[n] for n in self.names])
        zeros = lambda: OrderedDict((n, torch.zeros_like(v)) for n, v in self.main.items())
        self.m, self.v = zeros(), zeros()
        one = torch.tensor(1.0, dtype=dtype)
        self.b1p = {"pi": one * cfg.beta1, "q": one * cfg.beta1}
        self.b2p = {"pi": one * cfg.beta2, "q": one * cfg.beta2}
        self.grads = None

    def forward_losses(self, batch, eps_x, eps_x2=None, eps_t=None, main=None, frozen=None):
        """frozen = (q_backup, v_backup) evaluates the losses against fixed regression targets (what
        tf.stop_gradient means for a finite-difference check: v_backup depends on the main Q variables)."""
        cfg = self.cfg
        p = self.main if main is None else main
        x, x2, a = self._t(batch["obs1"]), self._t(batch["obs2"]), self._t(batch["acts"])
        r, d = self._t(batch["rews"]), self._t(batch["done"])
        mu, pi, logp_pi = policy(p, "main", x, self._t(eps_x), cfg, self.stable)
        q1, q2 = qf(p, "main/q1", x, a), qf(p, "main/q2", x, a)
        q1_pi, q2_pi = qf(p, "main/q1", x, pi), qf(p, "main/q2", x, pi)
        v = vf(p, "main/v", x)
        v_targ = vf(self.target, "target/v", x2)
        min_q_pi = torch.minimum(q1_pi, q2_pi)
        q_backup = (r + cfg.gamma * (1 - d) * v_targ).detach()
        v_backup = (min_q_pi - cfg.alpha * logp_pi).detach()
        if frozen is not None:
            q_backup, v_backup = frozen
        pi_loss = (cfg.alpha * logp_pi - q1_pi).mean()
        q1_loss = 0.5 * ((q_backup - q1) ** 2).mean()
        q2_loss = 0.5 * ((q_backup - q2) ** 2).mean()
        v_loss = 0.5 * ((v_backup - v) ** 2).mean()
        return dict(pi_loss=pi_loss, q1_loss=q1_loss, q2_loss=q2_loss, v_loss=v_loss, q1=q1, q2=q2, v=v, logp_pi=logp_pi,
                    pi=pi, mu=mu, q1_pi=q1_pi, q2_pi=q2_pi, q_backup=q_backup, v_backup=v_backup)

    def _value_names(self):
        return [n for n in self.names if "/q1/" in n or "/q2/" in n or "/v/" in n]

    def compute_grads(self, batch, eps_x, eps_x2=None, eps_t=None):
        leaves = OrderedDict((n, v.clone().requires_grad_(True)) for n, v in self.main.items())
        out = self.forward_losses(batch, eps_x, main=leaves)
        pi_names = [n for n in self.names if "/pi/" in n]
        v_names = self._value_names()
        g_pi = torch.autograd.grad(out["pi_loss"], [leaves[n] for n in pi_names], retain_graph=True)
        g_v = torch.autograd.grad(out["q1_loss"] + out["q2_loss"] + out["v_loss"], [leaves[n] for n in v_names])
        self.grads = OrderedDict(list(zip(pi_names, g_pi)) + list(zip(v_names, g_v)))
        return {k: (v.detach() if torch.is_tensor(v) else v) for k, v in out.items()}

    def apply_grads(self):
        c = self.cfg
        self._adam([n for n in self.names if "/pi/" in n], "pi")
        self._adam(self._value_names(), "q")
        pk = torch.tensor(c.polyak, dtype=self.dtype)
        pk1 = torch.tensor(1 - c.polyak, dtype=self.dtype)
        for n in self.names:
            tn = n.replace("main/", "target/", 1)
            self.target[tn] = pk * self.target[tn] + pk1 * self.main[n]

    def step(self, batch, eps_x, eps_x2=None, eps_t=None):
        out = self.compute_grads(batch, eps_x)
        self.apply_grads()
        return out
