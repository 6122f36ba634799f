"""ORACLE — TEST INFRASTRUCTURE ONLY.

CPU restatement (torch autograd) of the Double-DQN learner of algos/dqn/actor_learner.py:19-107 on
the network of algos/dqn/core.py:15-18,40-50.  Only tests/ may import this.

PARITY: composition PINNED by tests/golden/dqn_math.* and sqn_math.* — algos/dqn and algos/sqn actor_learner.py + core.py executed on
oracle/tf_shim.py (oracle/gen_golden_math.py), incl. config 5's learner shape (batch 512, 28 224-wide pixel observations);
tests/test_oracle_math_fixtures.py holds both oracles to them at 1e-10 in float64.  From memory (the shim's definitions): tf.one_hot /
tf.argmax (first maximum) / tf.stop_gradient / tf.nn.log_softmax / tf.make_template, tf.train.AdamOptimizer as in oracle/sac1_oracle.py.
"""
from collections import OrderedDict

import numpy as np
import torch


class Config:
    def __init__(self, obs_dim=8, n_actions=4, hidden1=400, hidden2=300, batch=128, gamma=0.99, lr=1e-3, polyak=0.995,
                 beta1=0.9, beta2=0.999, adam_eps=1e-8):
        self.__dict__.update(locals())
        del self.__dict__["self"]


def param_specs(cfg):
    return [("main/q1/dense/kernel", (cfg.obs_dim, cfg.hidden1)), ("main/q1/dense/bias", (cfg.hidden1,)),
            ("main/q1/dense_1/kernel", (cfg.hidden1, cfg.hidden2)), ("main/q1/dense_1/bias", (cfg.hidden2,)),
            ("main/q1/dense_2/kernel", (cfg.hidden2, cfg.n_actions)), ("main/q1/dense_2/bias", (cfg.n_actions,))]


def init_params(cfg, seed=0):
    rs = np.random.RandomState(seed)
    out = OrderedDict()
    for name, shape in param_specs(cfg):
        if name.endswith("kernel"):
            lim = np.sqrt(6.0 / (shape[0] + shape[1]))
            out[name] = rs.uniform(-lim, lim, size=shape).astype(np.float32)
        else:
            out[name] = np.zeros(shape, np.float32)
    return out


def q_net(p, scope, x):
    h = torch.relu(x @ p[scope + "/q1/dense/kernel"] + p[scope + "/q1/dense/bias"])
    h = torch.relu(h @ p[scope + "/q1/dense_1/kernel"] + p[scope + "/q1/dense_1/bias"])
    return h @ p[scope + "/q1/dense_2/kernel"] + p[scope + "/q1/dense_2/bias"]


class DqnOracle:
    def __init__(self, cfg, params, dtype=torch.float32):
        self.cfg, self.dtype = cfg, dtype
        self.names = [n for n, _ in param_specs(cfg)]
        self.main = OrderedDict((n, torch.tensor(np.asarray(params[n]), dtype=dtype).clone()) for n in self.names)
        self.target = OrderedDict((n.replace("main/", "target/", 1), v.clone()) for n, v in self.main.items())
        self.m = OrderedDict((n, torch.zeros_like(v)) for n, v in self.main.items())
        self.v = OrderedDict((n, torch.zeros_like(v)) for n, v in self.main.items())
        one = torch.tensor(1.0, dtype=dtype)
        self.b1p, self.b2p = one * cfg.beta1, one * cfg.beta2
        self.grads = None

    def _t(self, x):
        return torch.as_tensor(np.asarray(x)).to(self.dtype)

    def forward_loss(self, batch, main=None, frozen=None):
        c = self.cfg
        p = self.main if main is None else main
        x, x2, r, d = self._t(batch["obs1"]), self._t(batch["obs2"]), self._t(batch["rews"]), self._t(batch["done"])
        a = torch.as_tensor(np.asarray(batch["acts"])).to(torch.int64)            # tf.cast(a_ph, tf.int32)
        q, q_x2 = q_net(p, "main", x), q_net(p, "main", x2)
        q_next = q_net(self.target, "target", x2)
        q_value = (q * torch.nn.functional.one_hot(a, c.n_actions).to(self.dtype)).sum(1)
        best = torch.argmax(q_x2, dim=1)                                          # DDQN: online argmax, target value
        q_target = q_next.gather(1, best[:, None]).squeeze(1)
        q_backup = (r + c.gamma * (1 - d) * q_target).detach() if frozen is None else frozen
        q_loss = 0.5 * ((q_backup - q_value) ** 2).mean()
        return dict(q_loss=q_loss, q=q, q_backup=q_backup)

    def step(self, batch):
        c = self.cfg
        leaves = OrderedDict((n, v.clone().requires_grad_(True)) for n, v in self.main.items())
        out = self.forward_loss(batch, main=leaves)
        g = torch.autograd.grad(out["q_loss"], list(leaves.values()))
        self.grads = OrderedDict(zip(self.names, g))
        one = torch.tensor(1.0, dtype=self.dtype)
        lr, b1, b2, eps = (torch.tensor(v, dtype=self.dtype) for v in (c.lr, c.beta1, c.beta2, c.adam_eps))
        alpha_t = lr * torch.sqrt(one - self.b2p) / (one - self.b1p)
        for n in self.names:
            gg = self.grads[n]
            self.m[n] = self.m[n] + (gg - self.m[n]) * (one - b1)
            self.v[n] = self.v[n] + (gg * gg - self.v[n]) * (one - b2)
            self.main[n] = self.main[n] - (self.m[n] * alpha_t) / (torch.sqrt(self.v[n]) + eps)
        self.b1p, self.b2p = self.b1p * b1, self.b2p * b2
        pk, pk1 = torch.tensor(c.polyak, dtype=self.dtype), torch.tensor(1 - c.polyak, dtype=self.dtype)
        for n in self.names:
            tn = n.replace("main/", "target/", 1)
            self.target[tn] = pk * self.target[tn] + pk1 * self.main[n]
        return {k: (v.detach() if torch.is_tensor(v) else v) for k, v in out.items()}

    def flat(self, which="main"):
        d = {"main": self.main, "target": self.target, "m": self.m, "v": self.v, "grads": self.grads}[which]
        return np.concatenate([v.detach().numpy().reshape(-1) for v in d.values()])


def synthetic_batch(cfg, seed=0):
    rs = np.random.RandomState(seed)
    n = cfg.batch
    return dict(obs1=rs.randn(n, cfg.obs_dim).astype(np.float32), obs2=rs.randn(n, cfg.obs_dim).astype(np.float32),
                acts=rs.randint(0, cfg.n_actions, n).astype(np.float32), rews=rs.randn(n).astype(np.float32),
                done=(rs.rand(n) < 0.05).astype(np.float32))


# ---- SQN (algos/sqn/actor_learner.py:19-78, algos/sqn/core.py:30-79) -----------------------------------
def sqn_param_specs(cfg):
    out = []
    for q in ("q1", "q2"):
        out += [(n.replace("/q1/", "/%s/" % q), s) for n, s in param_specs(cfg)]
    return out


def sqn_init_params(cfg, seed=0):
    rs = np.random.RandomState(seed)
    out = OrderedDict()
    for name, shape in sqn_param_specs(cfg):
        if name.endswith("kernel"):
            lim = np.sqrt(6.0 / (shape[0] + shape[1]))
            out[name] = rs.uniform(-lim, lim, size=shape).astype(np.float32)
        else:
            out[name] = np.zeros(shape, np.float32)
    return out


def _mlp(p, prefix, x):
    h = torch.relu(x @ p[prefix + "/dense/kernel"] + p[prefix + "/dense/bias"])
    h = torch.relu(h @ p[prefix + "/dense_1/kernel"] + p[prefix + "/dense_1/bias"])
    return h @ p[prefix + "/dense_2/kernel"] + p[prefix + "/dense_2/bias"]


class SqnOracle(DqnOracle):
    def __init__(self, cfg, params, alpha=0.1, dtype=torch.float32):
        self.cfg, self.dtype, self.alpha = cfg, dtype, alpha
        self.names = [n for n, _ in sqn_param_specs(cfg)]
        self.main = OrderedDict((n, torch.tensor(np.asarray(params[n]), dtype=dtype).clone()) for n in self.names)
        self.target = OrderedDict((n.replace("main/", "target/", 1), v.clone()) for n, v in self.main.items())
        self.m = OrderedDict((n, torch.zeros_like(v)) for n, v in self.main.items())
        self.v = OrderedDict((n, torch.zeros_like(v)) for n, v in self.main.items())
        one = torch.tensor(1.0, dtype=dtype)
        self.b1p, self.b2p = one * cfg.beta1, one * cfg.beta2
        self.grads = None

    def forward_loss(self, batch, main=None, frozen=None):
        c = self.cfg
        p = self.main if main is None else main
        x, x2, r, d = self._t(batch["obs1"]), self._t(batch["obs2"]), self._t(batch["rews"]), self._t(batch["done"])
        a = torch.as_tensor(np.asarray(batch["acts"])).to(torch.int64)
        q1, q1_x2, q2 = _mlp(p, "main/q1", x), _mlp(p, "main/q1", x2), _mlp(p, "main/q2", x)
        pi_log = torch.log_softmax(q1_x2 / self.alpha, dim=1)
        entropy_x2 = (torch.exp(pi_log) * pi_log).sum(1)                     # core.py:41 ("exact entropy", sign as in the source)
        q1_t, q2_t = _mlp(self.target, "target/q1", x2), _mlp(self.target, "target/q2", x2)
        q1_mu_, q2_mu_ = q1_t.max(1).values, q2_t.max(1).values              # q * one_hot(argmax log_softmax(q / alpha))
        one_hot = torch.nn.functional.one_hot(a, c.n_actions).to(self.dtype)
        q1_a, q2_a = (q1 * one_hot).sum(1), (q2 * one_hot).sum(1)
        v_backup = (torch.minimum(q1_mu_, q2_mu_) - self.alpha * entropy_x2).detach()
        q_backup = r + c.gamma * (1 - d) * v_backup if frozen is None else frozen
        q_loss = 0.5 * ((q_backup - q1_a) ** 2).mean() + 0.5 * ((q_backup - q2_a) ** 2).mean()
        return dict(q_loss=q_loss, q=q1, q2=q2, q_backup=q_backup)
