"""Double-DQN learner / actor of the multi-node driver (algos/dqn/actor_learner.py:19-107, 175-201;
network algos/dqn/core.py:40-50) on the MI355X: `Learner(opt, job)` with set_weights / get_weights /
train(batch, cnt), `Actor(opt, job)` with get_action(o, deterministic) — same names and argument meaning.
`opt` carries obs_dim, act_dim (number of discrete actions), hidden_size, gamma, lr, polyak, batch_size, seed
(algos/dqn/hyperparams.py:26-60)."""
import ctypes
import math

import numpy as np
import torch

from . import _lib


class HyperParameters:
    """Value bag of algos/dqn/hyperparams.py:10-75 (algos/sqn/hyperparams.py adds `alpha`): what the DQN / SQN driver, buffers, server and
    agents read.  `env` (anything with observation_space.shape and action_space.n) or explicit obs_dim / act_dim."""

    def __init__(self, env=None, env_name="Trading", exp_name="ddqn-trading", num_nodes=1, num_workers=6, a_l_ratio=10, weights_file="",
                 obs_dim=None, act_dim=None):
        import datetime
        import os
        self.exp_name, self.env_name, self.model = exp_name, env_name, "mlp"
        self.num_nodes, self.num_workers, self.num_learners = num_nodes, num_workers, 1
        self.push_freq = 100
        self.gamma = 0.99
        self.alpha = 0.1                  # sqn: the softmax policy's temperature (must be > 0)
        self.a_l_ratio, self.weights_file = a_l_ratio, weights_file
        self.recover = False
        self.checkpoint_freq = 21600      # seconds (6 h)
        self.hidden_size = [400, 300]
        self.obs_dim = int(env.observation_space.shape[0] if obs_dim is None else obs_dim)
        self.act_dim = int(env.action_space.n if act_dim is None else act_dim)
        self.obs_shape, self.act_shape = (self.obs_dim,), ()
        self.num_buffers = self.num_workers // 25 + 1
        self.buffer_size = int(1e6) // self.num_buffers
        self.start_steps = int(1e4) // self.num_buffers
        if self.weights_file:
            self.start_steps = self.buffer_size
        self.lr, self.polyak, self.batch_size = 1e-3, 0.995, 128
        self.Ln, self.save_freq, self.seed = 1, 1, 0
        root = os.getcwd()                # (the reference: the directory above its own source file)
        self.summary_dir, self.save_dir, self.save_interval = root + "/tboard_ray", root + "/" + exp_name, int(5e5)
        self.log_dir = "%s/%s-workers_num:%s%%%s%s-%s" % (self.summary_dir, datetime.datetime.now(), num_workers, a_l_ratio, env_name, exp_name)


def param_specs(obs_dim, n_actions, hidden_size, nets=("q1",)):
    """(name, shape) in TF variable-creation order: tf.make_template('q1', vf_mlp) [, 'q2'] under scope 'main'."""
    h1, h2 = hidden_size
    specs = []
    for q in nets:
        specs += [("main/%s/dense/kernel" % q, (obs_dim, h1)), ("main/%s/dense/bias" % q, (h1,)),
                  ("main/%s/dense_1/kernel" % q, (h1, h2)), ("main/%s/dense_1/bias" % q, (h2,)),
                  ("main/%s/dense_2/kernel" % q, (h2, n_actions)), ("main/%s/dense_2/bias" % q, (n_actions,))]
    return specs


def glorot_init(specs, seed):
    rs = np.random.RandomState(seed)
    parts = []
    for name, shape in specs:
        if name.endswith("kernel"):
            lim = math.sqrt(6.0 / (shape[0] + shape[1]))
            parts.append(rs.uniform(-lim, lim, size=shape).astype(np.float32).reshape(-1))
        else:
            parts.append(np.zeros(int(np.prod(shape)), np.float32))
    return np.concatenate(parts)


class Learner:
    NETS = ("q1",)
    VARIANT = 0  # DDRL_DDQN

    def __init__(self, opt, job="learner", batch=None):
        _lib.require_gpu()
        self._lib = _lib.load()
        self.opt = opt
        self.device = torch.device("cuda", torch.cuda.current_device())
        hs = list(opt.hidden_size)
        assert len(hs) == 2, "two hidden layers (algos/dqn/hyperparams.py:37)"
        self.specs = param_specs(opt.obs_dim, opt.act_dim, hs, self.NETS)
        self.keys = [n for n, _ in self.specs]
        self.table, off = {}, 0
        for n, s in self.specs:
            cnt = int(np.prod(s))
            self.table[n] = (off, cnt, s)
            off += cnt
        self.n_params = off
        self.cfg = _lib.DqnConfig(opt.obs_dim, opt.act_dim, hs[0], hs[1], int(opt.batch_size if batch is None else batch),
                                  gamma=opt.gamma, lr=opt.lr, polyak=opt.polyak, variant=self.VARIANT,
                                  alpha=float(getattr(opt, "alpha", 0.1)))
        h = ctypes.c_void_p()
        _lib.check(self._lib.ddrl_dqn_create(ctypes.byref(h), self.device.index, ctypes.byref(self.cfg)))
        self._h = h
        self.loss = torch.zeros(1, dtype=torch.float32, device=self.device)
        self._flat_set(torch.from_numpy(glorot_init(self.specs, getattr(opt, "seed", 0))).to(self.device))

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.ddrl_dqn_destroy(h)

    def _flat_set(self, flat):
        flat = flat.to(device=self.device, dtype=torch.float32).contiguous()
        _lib.check(self._lib.ddrl_dqn_set_weights(self._h, _lib.dptr(flat), _lib.stream_ptr()))

    def export(self, which=_lib.SAC1_MAIN):
        flat = torch.empty(self.n_params, dtype=torch.float32, device=self.device)
        _lib.check(self._lib.ddrl_dqn_export(self._h, which, _lib.dptr(flat), _lib.stream_ptr()))
        return flat

    def import_(self, which, flat):
        flat = flat.to(device=self.device, dtype=torch.float32).contiguous()
        assert flat.numel() == self.n_params
        _lib.check(self._lib.ddrl_dqn_import(self._h, which, _lib.dptr(flat), _lib.stream_ptr()))

    def get_weights(self):
        flat = self.export().cpu().numpy()
        return list(self.keys), [flat[o:o + n].reshape(s).copy() for (o, n, s) in (self.table[k] for k in self.keys)]

    def set_weights(self, variable_names, weights):
        # every variable given: nothing of the current parameters survives, so nothing is read back (a learner whose stream-K combine
        # timed out refuses to export — fresh parameters are exactly what it is waiting for)
        full = set(self.keys) <= set(variable_names)
        flat = torch.empty(self.n_params, dtype=torch.float32, device=self.device) if full else self.export()
        for k, w in zip(variable_names, weights):
            if k not in self.table:
                continue
            o, n, _ = self.table[k]
            w = w if torch.is_tensor(w) else torch.from_numpy(np.asarray(w, np.float32))
            flat[o:o + n] = w.to(self.device).reshape(-1)
        self._flat_set(flat)

    def _dev(self, x, shape):
        t = x if torch.is_tensor(x) else torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))
        return t.to(device=self.device, dtype=torch.float32).contiguous().reshape(shape)

    def train(self, batch, cnt=0, return_outputs=False):
        """sess.run([q_loss, q, train_value_op, target_update]) (actor_learner.py:110-119)."""
        B = self.cfg.batch
        x, x2 = self._dev(batch["obs1"], (B, -1)), self._dev(batch["obs2"], (B, -1))
        a, r, d = self._dev(batch["acts"], (B,)), self._dev(batch["rews"], (B,)), self._dev(batch["done"], (B,))
        q = torch.empty(B, self.cfg.n_actions, dtype=torch.float32, device=self.device) if return_outputs else None
        _lib.check(self._lib.ddrl_dqn_step(self._h, _lib.dptr(x), _lib.dptr(x2), _lib.dptr(a), _lib.dptr(r), _lib.dptr(d),
                                           _lib.dptr(self.loss), _lib.dptr(q), _lib.stream_ptr()))
        if return_outputs:
            return self.loss, q
        return None

    def train_from(self, replay_buffer, cnt=0, return_outputs=False, with_indices=False):
        """One iteration of the learner's loop — `agent.train(replay_buffer.sample_batch(), cnt)` (algos/dqn/train.py:66-76) — with the
        layer-1 forward reading its observation rows straight out of the device ring (ddrl_dqn_step_ring): same index stream, same
        counters, bit-identical results, a quarter of the batch materialised.  Falls back to sample_batch_device + train where the fused
        path does not apply (a compact ring, observations narrower than 1024)."""
        B = self.cfg.batch
        q = torch.empty(B, self.cfg.n_actions, dtype=torch.float32, device=self.device) if return_outputs else None
        idx = torch.empty(B, dtype=torch.int64, device=self.device) if with_indices else None
        rc = self._lib.ddrl_dqn_step_ring(self._h, replay_buffer._h, _lib.dptr(self.loss), _lib.dptr(q), _lib.dptr(idx), _lib.stream_ptr())
        if rc == _lib.DDRL_ERR_UNSUPPORTED:
            b = replay_buffer.sample_batch_device(B, with_indices=with_indices)
            out = self.train(b, cnt, return_outputs=return_outputs)
            return (out + (b["idxs"],)) if (return_outputs and with_indices) else out
        _lib.check(rc)
        if return_outputs:
            return (self.loss, q, idx) if with_indices else (self.loss, q)
        return None

    def stage_times(self, batch, reps=10):
        """Mean milliseconds per launch group of `reps` updates on `batch` (ddrl_dqn_step_timed; bench.py's config-5 roofline)."""
        B = self.cfg.batch
        x, x2 = self._dev(batch["obs1"], (B, -1)), self._dev(batch["obs2"], (B, -1))
        a, r, d = self._dev(batch["acts"], (B,)), self._dev(batch["rews"], (B,)), self._dev(batch["done"], (B,))
        ms = (ctypes.c_float * _lib.DQN_STAGES)()
        _lib.check(self._lib.ddrl_dqn_step_timed(self._h, _lib.dptr(x), _lib.dptr(x2), _lib.dptr(a), _lib.dptr(r), _lib.dptr(d), int(reps), ms,
                                                 _lib.stream_ptr()))
        return [float(v) for v in ms]

    def _q_row(self, o):
        """q of ONE host observation as a NumPy row: page-locked staging both ways, one stream synchronisation."""
        st = getattr(self, "_stage_q", None)
        if st is None:
            hi, ho = torch.empty(1, self.cfg.obs_dim, dtype=torch.float32).pin_memory(), torch.empty(1, self.cfg.n_actions, dtype=torch.float32).pin_memory()
            # the q row is written straight into its page-locked row (device-side address), and a narrow observation is read straight
            # out of its own; pixel observations (the LDS-DMA layer-1 tiles) go up with a copy first
            po, pi = ctypes.c_void_p(), ctypes.c_void_p()
            _lib.check(self._lib.ddrl_host_device_pointer(ctypes.c_void_p(ho.data_ptr()), ctypes.byref(po)))
            di = None
            if self.cfg.obs_dim < 1024:
                _lib.check(self._lib.ddrl_host_device_pointer(ctypes.c_void_p(hi.data_ptr()), ctypes.byref(pi)))
            else:
                di = torch.empty(1, self.cfg.obs_dim, dtype=torch.float32, device=self.device)
            st = self._stage_q = (hi, hi.numpy(), di, pi, ho.numpy(), po, ho)
        hi, hiv, di, pi, hov, po, _ = st
        hiv[0, :] = np.asarray(o, np.float32).reshape(-1)
        if di is not None:
            di.copy_(hi, non_blocking=True)
        _lib.check(self._lib.ddrl_dqn_q(self._h, pi if di is None else _lib.dptr(di), 1, po, _lib.stream_ptr()))
        torch.cuda.current_stream().synchronize()
        return hov[0]

    def q_values(self, obs):
        obs = self._dev(obs, (-1, self.cfg.obs_dim))
        q = torch.empty(obs.shape[0], self.cfg.n_actions, dtype=torch.float32, device=self.device)
        _lib.check(self._lib.ddrl_dqn_q(self._h, _lib.dptr(obs), obs.shape[0], _lib.dptr(q), _lib.stream_ptr()))
        return q


class Actor(Learner):
    """algos/dqn/actor_learner.py:152-201: the q network alone; get_action(o) = argmax q with probability 0.97,
    a uniform random action otherwise (actor_learner.py:193-201; np.random there, a seeded RandomState here)."""

    def __init__(self, opt, job="worker", max_rows=1):
        super().__init__(opt, job, batch=max_rows)
        self._rs = np.random.RandomState(getattr(opt, "seed", 0))

    def get_action(self, o):
        if self._rs.uniform() < 0.97:
            return int(np.argmax(self._q_row(o)))
        return int(self._rs.randint(0, self.opt.act_dim))

    def _test_action(self, o):
        return self.get_action(o)

    def test(self, test_env, n=10):
        """actor_learner.py:230-251: n episodes to their terminal with the actor's own get_action; (mean return, mean of
        test_env.rewards[0] at each episode's end — the trading env's score)."""
        test_rets, scores = [], []
        for _ in range(n):
            o, r, d, ep_ret, ep_len = test_env.reset(), 0, False, 0, 0
            while True:
                o, r, d, _ = test_env.step(self._test_action(o))
                ep_ret += r
                ep_len += 1
                if d:
                    test_rets.append(ep_ret)
                    scores.append(test_env.rewards[0])
                    break
        return np.mean(test_rets), np.mean(scores)

    def write_tb(self, ave_test_reward, ave_score, alratio, update_frequency, total_learner_step):
        """actor_learner.py:203-228: the four test scalars ("Reward", "score", "a_l_ratio", "update_frequency") at step
        total_learner_step, into the run directory the reference names for job == "test" (actor_learner.py:173-176)."""
        if getattr(self, "_writer", None) is None:
            import datetime
            from .logx import SummaryWriter
            o = self.opt
            self._writer = SummaryWriter("%s/%s-%s-%s-workers_num:%s%%%s" % (getattr(o, "summary_dir", "."), datetime.datetime.now(), getattr(o, "env_name", ""),
                                                                            getattr(o, "exp_name", ""), getattr(o, "num_workers", 1), getattr(o, "a_l_ratio", "")))
        for tag, v in (("Reward", ave_test_reward), ("score", ave_score), ("a_l_ratio", alratio), ("update_frequency", update_frequency)):
            self._writer.add_scalar(tag, float(v), int(total_learner_step))
        self._writer.flush()


class LearnerSQN(Learner):
    """algos/sqn/actor_learner.py:19-131 (twin soft-Q networks main/q1, main/q2; step_ops = [q_loss, q1, q2, ...]):
    same surface as Learner; `opt.alpha` is the softmax policy's temperature (hyperparams.py:27)."""
    NETS = ("q1", "q2")
    VARIANT = 1  # DDRL_SQN


class ActorSQN(LearnerSQN):
    """algos/sqn/actor_learner.py:134-201: get_action(o, deterministic) = argmax of / a sample from
    softmax(q1(o) / alpha) (core.py:30-42; tf.random.multinomial there, a seeded RandomState here)."""

    def __init__(self, opt, job="worker", max_rows=1):
        super().__init__(opt, job, batch=max_rows)
        self._rs = np.random.RandomState(getattr(opt, "seed", 0))

    def get_action(self, o, deterministic=False):
        q = self._q_row(o).astype(np.float64)
        if deterministic:
            return int(np.argmax(q))
        z = q / float(self.opt.alpha)
        p = np.exp(z - z.max())
        return int(self._rs.choice(len(p), p=p / p.sum()))

    def _test_action(self, o):
        return self.get_action(o, deterministic=True)     # algos/sqn/actor_learner.py:238

    test, write_tb = Actor.test, Actor.write_tb
