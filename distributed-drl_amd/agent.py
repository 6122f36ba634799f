"""SAC1 agent classes with the reference's surface, backed by the HIP learner/actor kernels.

Mirrors (same names, argument meaning):
  algos/sac1/actor_learner.py:19-148   class Learner(opt, job): set_weights / get_weights / train
  algos/sac1/actor_learner.py:151-229  class Actor(opt, job):   set_weights / get_weights / get_action / test
  algos/sac1/hyperparams.py:10-104     class HyperParameters    (values; scalar obs_dim / act_dim,
                                        SURVEY §2.4 decision: alpha fixed at 0.1, action scale = high[0])

`get_weights()` returns `(keys, values)` restricted to names containing "main", keys being the TF1
variable names of the reference graph; `set_weights(keys, values)` accepts any subset.
tf.random_normal (core.py:77) is replaced by explicit noise: `train(batch, eps=...)` /
`get_action(o, eps=...)`; when omitted it is drawn on the device from the counter-based generator
(`ddrl_normal_fill`) seeded with `opt.seed`.
"""
import ctypes
import math
import time

import numpy as np
import torch

from . import _lib


def param_specs(obs_dim, act_dim, hidden1, hidden2, nets=("pi", "q1", "q2")):
    """(name, shape) in TF variable-creation order (core.py:91-121)."""
    o, a, h1, h2 = obs_dim, act_dim, hidden1, hidden2
    specs = []
    if "pi" in nets:
        specs += [("main/pi/dense/kernel", (o, h1)), ("main/pi/dense/bias", (h1,)),
                  ("main/pi/dense_1/kernel", (h1, h2)), ("main/pi/dense_1/bias", (h2,)),
                  ("main/pi/dense_2/kernel", (h2, a)), ("main/pi/dense_2/bias", (a,)),
                  ("main/pi/dense_3/kernel", (h2, a)), ("main/pi/dense_3/bias", (a,))]
    for q in ("q1", "q2"):
        if q in nets:
            specs += [("main/%s/dense/kernel" % q, (o + a, h1)), ("main/%s/dense/bias" % q, (h1,)),
                      ("main/%s/dense_1/kernel" % q, (h1, h2)), ("main/%s/dense_1/bias" % q, (h2,)),
                      ("main/%s/dense_2/kernel" % q, (h2, 1)), ("main/%s/dense_2/bias" % q, (1,))]
    if "v" in nets:  # example/core.py:112-113,126-127: vf_mlp(x)
        specs += [("main/v/dense/kernel", (o, h1)), ("main/v/dense/bias", (h1,)),
                  ("main/v/dense_1/kernel", (h1, h2)), ("main/v/dense_1/bias", (h2,)),
                  ("main/v/dense_2/kernel", (h2, 1)), ("main/v/dense_2/bias", (1,))]
    return specs


def glorot_init(specs, seed):
    """tf.layers.dense defaults: glorot-uniform kernels, zero biases; one flat float32 vector."""
    rs = np.random.RandomState(seed)
    parts = []
    for name, shape in specs:
        if name.endswith("kernel"):
            lim = math.sqrt(6.0 / (shape[0] + shape[1]))
            parts.append(rs.uniform(-lim, lim, size=shape).astype(np.float32).reshape(-1))
        else:
            parts.append(np.zeros(int(np.prod(shape)), np.float32))
    return np.concatenate(parts)


class HyperParameters:
    """Value bag of algos/sac1/hyperparams.py:10-104 (the entries the hot path reads)."""

    def __init__(self, env_name="LunarLanderContinuous-v2", exp_name="sac1", num_workers=1, a_l_ratio=2,
                 weights_file="", obs_dim=8, act_dim=2, act_scale=1.0):
        self.exp_name, self.env_name = exp_name, env_name
        self.a_l_ratio = a_l_ratio
        self.weights_file = weights_file
        self.start_steps = int(5e4) if not weights_file else int(10e6)
        self.obs_dim, self.act_dim, self.act_scale = obs_dim, act_dim, act_scale
        self.hidden_sizes = (400, 300)  # core.py:91
        self.num_workers, self.num_learners = num_workers, 1
        self.alpha = 0.1
        self.gamma = 0.997
        self.num_buffers = self.num_workers // 25 + 1
        self.buffer_size = int(3e6) // self.num_buffers
        self.lr = 5e-5
        self.polyak = 0.995
        self.steps_per_epoch = 5000
        self.batch_size = 256
        self.max_ep_len = 2900
        self.seed = 0
        self.push_freq = 300  # sac1.py:149
        # n-step driver (algos/sac1/sac_ray.py) entries: hyperparams.py:40-42, 84-88
        self.obs_noise, self.act_noise, self.reward_scale = 0, 0.3, 5
        self.Ln, self.action_repeat, self.save_freq = 8, 2, 1
        self.obs_shape, self.act_shape = (obs_dim,), (act_dim,)
        self.num_envs = 1     # vectorised rollouts: envs stepped together by one worker
        import os
        self.summary_dir = os.getcwd() + "/tboard_ray"   # hyperparams.py:99 (TensorBoard scalars of the job == "main" actor)

    def config(self, batch=None, variant=0):
        return _lib.Sac1Config(obs_dim=self.obs_dim, act_dim=self.act_dim, hidden1=self.hidden_sizes[0],
                               hidden2=self.hidden_sizes[1], batch=self.batch_size if batch is None else batch, variant=variant,
                               alpha=self.alpha, gamma=self.gamma, lr=self.lr, polyak=self.polyak,
                               act_scale=self.act_scale)


class _Net:
    def _setup(self, opt, nets, job="", index=0):
        _lib.require_gpu()
        self._lib = _lib.load()
        self.opt = opt
        self.device = torch.device("cuda", torch.cuda.current_device())
        self.specs = param_specs(opt.obs_dim, opt.act_dim, opt.hidden_sizes[0], opt.hidden_sizes[1], nets)
        self.keys = [n for n, _ in self.specs]
        off = 0
        self.table = {}
        for n, s in self.specs:
            cnt = int(np.prod(s))
            self.table[n] = (off, cnt, s)
            off += cnt
        self.n_params = off
        # one noise stream per (seed, job, worker): workers / ranks that share opt.seed must not draw the same eps sequence
        import zlib
        self._noise_seed = (int(getattr(opt, "seed", 0)) * 2654435761 + zlib.crc32(str(job).encode()) + 97 * int(index)) & 0xFFFFFFFF
        self._noise_ctr = 0

    def _flat_get(self):
        raise NotImplementedError

    def _flat_set(self, flat):
        raise NotImplementedError

    def get_weights_flat(self):
        """The whole "main" vector as ONE device tensor (what ps.push_flat / RCCL broadcast move)."""
        return self._flat_get()

    def set_weights_flat(self, flat):
        self._flat_set(flat.to(device=self.device, dtype=torch.float32).contiguous())

    def get_weights(self):
        flat = self._flat_get().cpu().numpy()
        values = [flat[o:o + n].reshape(s).copy() for (o, n, s) in (self.table[k] for k in self.keys)]
        return list(self.keys), values

    def set_weights(self, variable_names, weights):
        if list(variable_names) == self.keys:
            flat = np.concatenate([np.asarray(w, np.float32).reshape(-1) if not torch.is_tensor(w)
                                   else w.detach().float().cpu().numpy().reshape(-1) for w in weights])
            self._flat_set(torch.from_numpy(flat).to(self.device))
            return
        flat = self._flat_get()
        for k, w in zip(variable_names, weights):
            if k not in self.table:
                continue  # the reference's TensorFlowVariables ignores names it does not hold
            o, n, _ = self.table[k]
            w = w if torch.is_tensor(w) else torch.from_numpy(np.asarray(w, np.float32))
            flat[o:o + n] = w.to(self.device).reshape(-1)
        self._flat_set(flat)

    def _normal(self, n):
        out = torch.empty(n, dtype=torch.float32, device=self.device)
        _lib.check(self._lib.ddrl_normal_fill(_lib.dptr(out), n, self._noise_seed, self._noise_ctr, _lib.stream_ptr()))
        self._noise_ctr += n
        return out

    def _dev(self, x, shape=None):
        t = x if torch.is_tensor(x) else torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))
        t = t.to(device=self.device, dtype=torch.float32).contiguous()
        return t if shape is None else t.reshape(shape)


class Learner(_Net):
    """algos/sac1/actor_learner.py:19-148."""

    NETS = ("pi", "q1", "q2")
    VARIANT = 0   # _lib.SAC1
    N_LOSSES = 3

    def __init__(self, opt, job="learner", index=0):
        self._setup(opt, self.NETS, job, index)
        self.cfg = opt.config(variant=self.VARIANT)
        h = ctypes.c_void_p()
        _lib.check(self._lib.ddrl_sac1_create(ctypes.byref(h), self.device.index, ctypes.byref(self.cfg)))
        self._h = h
        self.losses = torch.zeros(self.N_LOSSES, dtype=torch.float32, device=self.device)
        # tf.global_variables_initializer with tf.set_random_seed(opt.seed): glorot / zeros
        self._flat_set(torch.from_numpy(glorot_init(self.specs, opt.seed)).to(self.device))

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.ddrl_sac1_destroy(h)

    def _flat_get(self):
        flat = torch.empty(self.n_params, dtype=torch.float32, device=self.device)
        _lib.check(self._lib.ddrl_sac1_get_weights(self._h, _lib.dptr(flat), _lib.stream_ptr()))
        return flat

    def _flat_set(self, flat):
        # set_weights + target_init (actor_learner.py:125-127)
        _lib.check(self._lib.ddrl_sac1_set_weights(self._h, _lib.dptr(flat), _lib.stream_ptr()))

    def export(self, which, out=None):
        flat = torch.empty(self.n_params, dtype=torch.float32, device=self.device) if out is None else out
        _lib.check(self._lib.ddrl_sac1_export(self._h, which, _lib.dptr(flat), _lib.stream_ptr()))
        return flat

    def import_(self, which, flat):
        flat = flat.to(device=self.device, dtype=torch.float32).contiguous()
        _lib.check(self._lib.ddrl_sac1_import(self._h, which, _lib.dptr(flat), _lib.stream_ptr()))

    def opt_steps(self):
        a, b = ctypes.c_int64(), ctypes.c_int64()
        _lib.check(self._lib.ddrl_sac1_opt_steps(self._h, ctypes.byref(a), ctypes.byref(b), _lib.stream_ptr()))
        return int(a.value), int(b.value)

    def save_state(self, path):
        """Full learner checkpoint (own .npz format; the reference only pickles the weights): main / target
        parameters, Adam moments, step counts and the noise counter.  load_state() resumes bit-identically."""
        import ctypes
        t_pi, t_q, ctr = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_uint64()
        _lib.check(self._lib.ddrl_sac1_opt_state_get(self._h, ctypes.byref(t_pi), ctypes.byref(t_q), ctypes.byref(ctr), _lib.stream_ptr()))
        np.savez(path, main=self.export(_lib.SAC1_MAIN).cpu().numpy(), target=self.export(_lib.SAC1_TARGET).cpu().numpy(),
                 adam_m=self.export(_lib.SAC1_ADAM_M).cpu().numpy(), adam_v=self.export(_lib.SAC1_ADAM_V).cpu().numpy(),
                 opt=np.array([t_pi.value, t_q.value, ctr.value], dtype=np.uint64), keys=np.array(self.keys))

    def load_state(self, path):
        z = np.load(path if str(path).endswith(".npz") else str(path) + ".npz")
        assert list(z["keys"]) == list(self.keys), "checkpoint is for a different network"
        for which, name in ((_lib.SAC1_MAIN, "main"), (_lib.SAC1_TARGET, "target"), (_lib.SAC1_ADAM_M, "adam_m"), (_lib.SAC1_ADAM_V, "adam_v")):
            self.import_(which, torch.from_numpy(z[name]).to(self.device))
        t_pi, t_q, ctr = (int(x) for x in z["opt"])
        _lib.check(self._lib.ddrl_sac1_opt_state_set(self._h, t_pi, t_q, ctr, _lib.stream_ptr()))

    def _host_fast(self, batch):
        """train(replay_buffer.sample_batch()) — the reference's learner loop body with HOST arrays (example/dsac.py:142-144) — without any
        staging launch: the five arrays go up as ONE asynchronous copy out of a page-locked block (two of them, alternating behind events)
        straight INTO the learner's input set 0, the three noise tensors are generated in place behind it, and the update reads the input
        set it owns — copy, noise and update issued by ONE C call (ddrl_sac1_step_host: on this surface the interpreter, not the device,
        sets the rate).  Same values, same noise stream positions as the general path.
        -> True when the update has been issued, None when the input set's items do not lie the way this needs."""
        B, o, a = self.cfg.batch, self.cfg.obs_dim, self.cfg.act_dim
        st = getattr(self, "_fast", None)
        if st is None:
            bufs = (ctypes.c_void_p * 8)()
            _lib.check(self._lib.ddrl_sac1_input_buffers(self._h, 0, bufs))
            p = [int(bufs[i]) for i in range(8)]
            n = (B * o, B * o, B * a, B, B)
            off = [(p[j] - p[0]) // 4 for j in range(5)]
            # one block copy covers [obs1 | obs2 | acts | rews | done]: the five items must follow one another exactly as the learner
            # allocates them — rup32(batch) rows each on the direct path (the rows past the batch are zero padding and stay zero: the
            # page-locked block holds zeros there), rounded up to 64 floats — so that no other live buffer lies inside the span
            Bp = (B + 31) // 32 * 32 if self._lib.ddrl_sac1_is_fused(self._h) == 1 else B
            alloc = [(Bp * w + 63) // 64 * 64 for w in (o, o, a, 1, 1)]
            ok = all((p[j] - p[0]) % 4 == 0 for j in range(5)) and all(off[j + 1] - off[j] == alloc[j] for j in range(4))
            if not ok:
                st = self._fast = False
            else:
                span = off[4] + B
                hosts = [torch.zeros(span + 2, dtype=torch.float32).pin_memory() for _ in range(2)]   # (+2: the noise counter rides behind the batch)
                st = self._fast = {"p": p, "off": off, "n": n, "span": span, "host": hosts, "np": [h.numpy() for h in hosts],
                                   "hp": [ctypes.c_void_p(h.data_ptr()) for h in hosts],
                                   "ev": [torch.cuda.Event(), torch.cuda.Event()], "used": [False, False], "turn": 0, "one_block": Bp == B,
                                   }
        if st is False:
            return None
        t = st["turn"]
        st["turn"] = t ^ 1
        if st["used"][t]:
            st["ev"][t].synchronize()                 # the copy that last read this staging block has left it
        hv, off, n = st["np"][t], st["off"], st["n"]
        arrs = [batch[k] for k in ("obs1", "obs2", "acts", "rews", "done")]
        p0 = arrs[0].ctypes.data
        if st["one_block"] and all(x.dtype == np.float32 and x.flags.c_contiguous and x.size == n[j] and x.ctypes.data - p0 == 4 * off[j] for j, x in enumerate(arrs)):
            # the five arrays are pieces of ONE host block with the input set's own offsets (what ReplayBuffer.sample_batch hands out
            # at the reference's shapes): one memmove instead of five conversions
            ctypes.memmove(st["hp"][t], p0, 4 * st["span"])
        else:
            for j, x in enumerate(arrs):
                hv[off[j]:off[j] + n[j]] = np.asarray(x, dtype=np.float32).reshape(n[j])
        _lib.check(self._lib.ddrl_sac1_step_host(self._h, st["hp"][t], st["span"], self._noise_seed, self._noise_ctr, _lib.dptr(self.losses),
                                                 _lib.stream_ptr()))
        st["ev"][t].record()
        st["used"][t] = True
        self._noise_ctr += 3 * B * a
        return True

    def _guard_stepper(self):
        """A caller's batch is staged into input set 0 (by copy kernel or by the one-block host copy): a batch the data-parallel
        stepper has drawn AHEAD into that set would be overwritten and its next update would silently train on this one."""
        dp = getattr(self, "_dp_state", None)
        if dp is not None and dp["primed"]:
            raise RuntimeError("train / compute_gradients with a caller's batch while a dp_stepper holds a batch drawn ahead: end its "
                               "sequence with grads(last=True) first")

    def _args(self, batch, eps, outs):
        B, a = self.cfg.batch, self.cfg.act_dim
        self._guard_stepper()
        if all(isinstance(batch[k], np.ndarray) for k in ("obs1", "obs2", "acts", "rews", "done")):
            if eps is None and not outs and getattr(self, "_fast_step", False):
                if self._host_fast(batch):
                    return None, None, (None, None, None)     # (the one call has issued the update)
            x, x2, ac, r, d = self._host_batch(batch)
        else:
            x, x2 = self._dev(batch["obs1"], (B, -1)), self._dev(batch["obs2"], (B, -1))
            ac, r, d = self._dev(batch["acts"], (B, a)), self._dev(batch["rews"], (B,)), self._dev(batch["done"], (B,))
        if eps is None:
            e = self._normal(3 * B * a).view(3, B, a)
            eps = (e[0], e[1], e[2])
        eps = [self._dev(e, (B, a)) for e in eps]
        keep = (x, x2, ac, r, d, *eps)
        q1 = q2 = lp = None
        if outs:
            q1, q2, lp = (torch.empty(B, dtype=torch.float32, device=self.device) for _ in range(3))
        ptrs = [_lib.dptr(t) for t in keep] + [_lib.dptr(self.losses), _lib.dptr(q1), _lib.dptr(q2), _lib.dptr(lp)]
        return keep, ptrs, (q1, q2, lp)

    def _host_batch(self, batch):
        """The reference's feed (five NumPy arrays) crosses PCIe as ONE block out of a page-locked staging buffer (the copy is
        synchronous: the buffer is free again on return; the device block is reused stream-ordered behind the previous update)."""
        B, o, a = self.cfg.batch, self.cfg.obs_dim, self.cfg.act_dim
        st = getattr(self, "_stage_b", None)
        if st is None:
            offs = [0]
            for n in (B * o, B * o, B * a, B, B):
                offs.append((offs[-1] + n + 3) & ~3)
            host = torch.empty(offs[5], dtype=torch.float32).pin_memory()
            st = self._stage_b = (host, host.numpy(), torch.empty(offs[5], dtype=torch.float32, device=self.device), offs)
        host, hv, dev, offs = st
        for j, (k, n) in enumerate((("obs1", B * o), ("obs2", B * o), ("acts", B * a), ("rews", B), ("done", B))):
            hv[offs[j]:offs[j] + n] = np.asarray(batch[k], dtype=np.float32).reshape(n)
        dev.copy_(host)
        return (dev[offs[0]:offs[0] + B * o].view(B, o), dev[offs[1]:offs[1] + B * o].view(B, o), dev[offs[2]:offs[2] + B * a].view(B, a),
                dev[offs[3]:offs[3] + B], dev[offs[4]:offs[4] + B])

    def train(self, batch, eps=None, return_outputs=False):
        """One sess.run(step_ops) (actor_learner.py:135-142).  `batch` holds obs1/obs2/acts/rews/done
        as NumPy arrays (the reference's feed) or device tensors (no host round trip)."""
        self._fast_step = True                    # train(): a host batch may take the one-call path (copy + noise + update)
        try:
            keep, ptrs, outs = self._args(batch, eps, return_outputs)
        finally:
            self._fast_step = False
        if ptrs is not None:
            _lib.check(self._lib.ddrl_sac1_step(self._h, *ptrs, _lib.stream_ptr()))
        if return_outputs:
            return self.losses, outs
        return None  # the reference drops the fetched values (actor_learner.py:142)

    def _device_args(self, batch):
        """Pointers for a batch that already lives on the device, with the noise generated inside the update's first
        kernel from the learner's device counter (ddrl_sac1_fill_noise): no noise launch, no host round trip."""
        if getattr(self, "_in0", None) is None:
            bufs = (ctypes.c_void_p * 8)()
            _lib.check(self._lib.ddrl_sac1_input_buffers(self._h, 0, bufs))
            self._in0 = [bufs[i] for i in range(8)]
        self._guard_stepper()
        B, a = self.cfg.batch, self.cfg.act_dim
        keep = [self._dev(batch[k], s) for k, s in (("obs1", (B, -1)), ("obs2", (B, -1)), ("acts", (B, a)), ("rews", (B,)), ("done", (B,)))]
        _lib.check(self._lib.ddrl_sac1_fill_noise(self._h, self._noise_seed, _lib.stream_ptr()))
        return keep, [_lib.dptr(t) for t in keep] + self._in0[5:8] + [None, None, None, None]

    def input_batch(self, which=0):
        """Views of input set `which` (0/1) of the learner — where the device loop's sampler gathers the batch of every
        other update (ddrl_sac1_input_buffers): obs1, obs2, acts, rews, done."""
        from .replay import _view
        bufs = (ctypes.c_void_p * 8)()
        _lib.check(self._lib.ddrl_sac1_input_buffers(self._h, int(which), bufs))
        B, o, a = self.cfg.batch, self.cfg.obs_dim, self.cfg.act_dim
        dev = torch.device("cuda", torch.cuda.current_device())
        return {k: _view(bufs[i], s, dev) for i, (k, s) in enumerate((("obs1", (B, o)), ("obs2", (B, o)), ("acts", (B, a)), ("rews", (B,)), ("done", (B,))))}

    def train_device(self, batch):
        """train() for a device-resident batch with in-kernel noise (what the graph loop does per update, issued eagerly)."""
        keep, ptrs = self._device_args(batch)
        _lib.check(self._lib.ddrl_sac1_step(self._h, *ptrs, _lib.stream_ptr()))

    def compute_gradients_device(self, batch, out=None):
        keep, ptrs = self._device_args(batch)
        _lib.check(self._lib.ddrl_sac1_compute_grads(self._h, *ptrs, _lib.stream_ptr()))
        return self.export(_lib.SAC1_GRAD, out)

    def dp_stepper(self, ring):
        """Data-parallel learner iteration with the host work done once (partition.py, config 4): returns (grads, apply, g) —
        grads() runs forward + backward on the batch `ring`'s sampler (its feed plan included) drew into one of the two input
        sets — the draw of the following batch rides in that update's forward launch — leaving the COMPLETE gradient in the learner's own buffer `g` (a view, internal layout) for an in-place
        all-reduce; apply() steps Adam + polyak with it.  No staging, export or import copies, no per-call argument building."""
        from .replay import _view
        lib, sp = self._lib, _lib.stream_ptr
        ins = []
        for st in range(2):
            bufs = (ctypes.c_void_p * 8)()
            _lib.check(lib.ddrl_sac1_input_buffers(self._h, st, bufs))
            ins.append([ctypes.c_void_p(bufs[i]) for i in range(8)])
        gp, gn = ctypes.c_void_p(), ctypes.c_int64()
        _lib.check(lib.ddrl_sac1_grad_buffer(self._h, ctypes.byref(gp), ctypes.byref(gn)))
        g = _view(gp.value, (int(gn.value),), torch.device("cuda", torch.cuda.current_device()))
        B, h, rh, seed, nul = int(self.cfg.batch), self._h, ring._h, self._noise_seed, ctypes.c_void_p(None)
        state = self._dp_state = {"cur": 0, "primed": False}   # (train(host batch) looks at `primed` before it writes into input set 0)

        def grads(last=False):
            """`last`: the final update before the ring / its feed plan changes (end of a step): nothing is drawn ahead."""
            s, cur = sp(), state["cur"]
            if not state["primed"]:
                i = ins[cur]
                _lib.check(lib.ddrl_replay_sample(rh, B, i[0], i[1], i[2], i[3], i[4], nul, s))
            _lib.check(lib.ddrl_sac1_fill_noise(h, seed, s))
            if last:
                _lib.check(lib.ddrl_sac1_compute_grads(h, *ins[cur], nul, nul, nul, nul, s))
                state["primed"] = False
            else:   # the next batch's sampler rides in this update's forward launch, into the other input set
                _lib.check(lib.ddrl_sac1_compute_grads_and_sample(h, cur, rh, cur ^ 1, s))
                state["primed"], state["cur"] = True, cur ^ 1
            _lib.check(lib.ddrl_sac1_grad_finalize(h, s))
            self._dp_last_set = cur

        def apply():
            _lib.check(lib.ddrl_sac1_apply_grads(h, sp()))

        def graph_sync():
            """Before a caller's graph capture and as its last captured call: optimizer state / dgrad image on copy 0; the
            stepper itself restarts from input set 0 with nothing drawn ahead (a graph ends with a `last` update)."""
            assert not state["primed"], "a captured sequence must end with grads(last=True)"
            state["cur"] = 0
            _lib.check(lib.ddrl_sac1_graph_sync(h, sp()))
        grads.graph_sync = graph_sync

        def reset():
            """After an aborted graph capture (nothing it recorded has run): nothing drawn ahead, input set 0 next."""
            state["cur"], state["primed"] = 0, False
        grads.reset = reset
        return grads, apply, g

    def capture_begin(self):
        """Snapshot the handle's host-side launch state right before a stream capture of this learner's launches ..."""
        _lib.check(self._lib.ddrl_sac1_capture_begin(self._h))

    def capture_abort(self):
        """... and put it back when that capture was aborted (nothing it recorded has run): include/ddrl.h."""
        _lib.check(self._lib.ddrl_sac1_capture_abort(self._h))

    def compute_gradients(self, batch, eps=None):
        """Forward + backward only (the stubbed compute_gradients of actor_learner.py:144-145)."""
        keep, ptrs, _ = self._args(batch, eps, False)
        _lib.check(self._lib.ddrl_sac1_compute_grads(self._h, *ptrs, _lib.stream_ptr()))
        return self.export(_lib.SAC1_GRAD)

    def apply_gradients(self, gradients=None):
        """Adam(pi), Adam(q), polyak with `gradients` (flat) or the buffer of the last compute."""
        if gradients is not None:
            self.import_(_lib.SAC1_GRAD, gradients)
        _lib.check(self._lib.ddrl_sac1_apply_grads(self._h, _lib.stream_ptr()))


class _ModelOpt:
    """example/dsac.py:185-216's `args` as the option bag the kernels read."""

    def __init__(self, args):
        hs = list(args.ac_kwargs.get("hidden_sizes", (400, 300))) if hasattr(args, "ac_kwargs") else list(getattr(args, "hidden_sizes", (400, 300)))
        if len(hs) == 1:
            hs = hs * 2
        assert len(hs) == 2, "two hidden layers (example/dsac.py: --l 2; --l 1 is widened to two equal layers here)"
        self.hidden_sizes = tuple(int(h) for h in hs)
        self.obs_dim, self.act_dim = int(args.obs_dim), int(args.act_dim)
        g = args.gamma[0] if isinstance(args.gamma, (tuple, list)) else args.gamma   # dsac.py:198 sets a 1-tuple
        self.alpha, self.gamma, self.lr, self.polyak = float(args.alpha), float(g), float(args.lr), float(args.polyak)
        self.batch_size = int(args.batch_size)
        self.seed = int(getattr(args, "seed", 0))
        sp = args.ac_kwargs.get("action_space") if hasattr(args, "ac_kwargs") else None
        self.act_scale = float(sp.high[0]) if sp is not None else float(getattr(args, "act_scale", 1.0))
        self.max_ep_len = int(getattr(args, "max_ep_len", 1000))

    def config(self, batch=None, variant=0):
        return _lib.Sac1Config(obs_dim=self.obs_dim, act_dim=self.act_dim, hidden1=self.hidden_sizes[0], hidden2=self.hidden_sizes[1],
                               batch=self.batch_size if batch is None else batch, variant=variant, alpha=self.alpha,
                               gamma=self.gamma, lr=self.lr, polyak=self.polyak, act_scale=self.act_scale)


class Model(Learner):
    """example/model.py:12-118 — the SAC-v learner of example/dsac.py (policy + twin Q + V + target V):
    `Model(args)`, set_weights / get_weights ("main" variables: pi, q1, q2, v), get_action(o, deterministic),
    train(replay_buffer, args) = one sample_batch + one sess.run(step_ops), test_agent(test_env, args, n)."""
    NETS = ("pi", "q1", "q2", "v")
    VARIANT = 1   # _lib.SAC_V
    N_LOSSES = 4  # pi_loss, q1_loss, q2_loss, v_loss (model.py:66)

    def __init__(self, args):
        super().__init__(_ModelOpt(args) if not isinstance(args, (HyperParameters, _ModelOpt)) else args)
        self._actor = None

    def _policy(self):
        if self._actor is None:
            self._actor = Actor(self.opt, job="worker", max_rows=1)
        flat = self._flat_get()
        n = self._actor.n_params
        self._actor.set_weights_flat(flat[:n])   # the policy variables lead the flat vector
        return self._actor

    def get_action(self, o, deterministic=False):
        a = self._policy().get_actions(torch.as_tensor(np.asarray(o, dtype=np.float32)).reshape(1, -1), deterministic=deterministic)
        return a[0].cpu().numpy()

    def train(self, replay_buffer, args=None, eps=None, return_outputs=False):
        if isinstance(replay_buffer, dict):   # a batch, as Learner.train takes it
            batch = replay_buffer
        else:
            from .workers import _remote, _get
            batch = _get(_remote(replay_buffer.sample_batch, self.cfg.batch if args is None else args.batch_size))
        return super().train(batch, eps=eps, return_outputs=return_outputs)

    def test_agent(self, test_env, args, n=10):
        test_ret = []
        for _ in range(n):
            o, r, d, ep_ret, ep_len = test_env.reset(), 0, False, 0, 0
            while not (d or (ep_len == args.max_ep_len)):
                o, r, d, _ = test_env.step(self.get_action(o, True))
                ep_ret += r
                ep_len += 1
            test_ret.append(ep_ret)
        return sum(test_ret) / len(test_ret)


class Actor(_Net):
    """algos/sac1/actor_learner.py:151-229 (policy-only graph)."""

    def __init__(self, opt, job="worker", max_rows=None, index=0):
        self._setup(opt, ("pi",), job, index)
        self.cfg = opt.config()
        self.max_rows = int(max_rows if max_rows is not None else max(1, getattr(opt, "num_envs", 1)))
        h = ctypes.c_void_p()
        _lib.check(self._lib.ddrl_actor_create(ctypes.byref(h), self.device.index, ctypes.byref(self.cfg), self.max_rows))
        self._h = h
        self.job = job
        self._flat_set(torch.from_numpy(glorot_init(self.specs, opt.seed)).to(self.device))

    def __del__(self):
        w = getattr(self, "_writer", None)
        if w is not None and hasattr(w, "close"):
            w.close()
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.ddrl_actor_destroy(h)

    def _flat_get(self):
        flat = torch.empty(self.n_params, dtype=torch.float32, device=self.device)
        _lib.check(self._lib.ddrl_actor_get_weights(self._h, _lib.dptr(flat), _lib.stream_ptr()))
        return flat

    def _flat_set(self, flat):
        _lib.check(self._lib.ddrl_actor_set_weights(self._h, _lib.dptr(flat), _lib.stream_ptr()))

    # -- version store: exact per-env weight adoption of a vectorised rollout worker (example/dsac.py:127-130) ------------------
    def enable_versions(self, n_slots):
        """Keep `n_slots` resident copies of the policy: set_weights then stores a NEW version (instead of replacing the one every
        env acts on) and the fused rollout step moves an env to the newest version at that env's own episode end."""
        _lib.check(self._lib.ddrl_actor_versions_enable(self._h, int(n_slots), _lib.stream_ptr()))
        self.n_slots = int(n_slots)

    def version_state(self, with_slots=True):
        """-> (slot of every env [max_rows] int32 device tensor or None, dict(newest, live, tiles, out_of_slots))."""
        slots = torch.empty(self.max_rows, dtype=torch.int32, device=self.device) if with_slots else None
        st = (ctypes.c_int32 * 4)()
        _lib.check(self._lib.ddrl_actor_versions_state(self._h, _lib.dptr(slots), st, _lib.stream_ptr()))
        return slots, {"newest": int(st[0]), "live": int(st[1]), "tiles": int(st[2]), "out_of_slots": bool(st[3])}

    def adopt_where_ended(self, ended):
        """Episode ends of steps taken outside the fused rollout step (uint8 mask of env.step): those envs pull."""
        ended = ended.to(device=self.device, dtype=torch.uint8).contiguous()
        _lib.check(self._lib.ddrl_actor_versions_adopt(self._h, _lib.dptr(ended), int(ended.numel()), _lib.stream_ptr()))

    def get_actions_versioned(self, obs, horizon_steps, deterministic=False, eps=None, out=None):
        """get_actions with every env evaluated against the policy version in its slot (enable_versions first; all max_rows envs)."""
        obs = self._dev(obs, (-1, self.cfg.obs_dim))
        n, a = obs.shape[0], self.cfg.act_dim
        if not deterministic:
            eps = self._normal(n * a).view(n, a) if eps is None else self._dev(eps, (n, a))
        act = out if out is not None else torch.empty(n, a, dtype=torch.float32, device=self.device)
        _lib.check(self._lib.ddrl_actor_act_versioned(self._h, _lib.dptr(obs), _lib.dptr(eps if not deterministic else None), n,
                                                      1 if deterministic else 0, int(horizon_steps), _lib.dptr(act), _lib.stream_ptr()))
        return act

    def get_actions(self, obs, deterministic=False, eps=None, out=None):
        """Batched get_action on device tensors: obs[n, obs_dim] -> act[n, act_dim]."""
        obs = self._dev(obs, (-1, self.cfg.obs_dim))
        n, a = obs.shape[0], self.cfg.act_dim
        if not deterministic:
            eps = self._normal(n * a).view(n, a) if eps is None else self._dev(eps, (n, a))
        act = out if out is not None else torch.empty(n, a, dtype=torch.float32, device=self.device)
        _lib.check(self._lib.ddrl_actor_act(self._h, _lib.dptr(obs), _lib.dptr(eps if not deterministic else None), n,
                                            1 if deterministic else 0, _lib.dptr(act), _lib.stream_ptr()))
        return act

    def get_action(self, o, deterministic=False, eps=None):
        """actor_learner.py:195-197: one observation in, one action (NumPy) out.  The observation goes up and the action comes down
        through page-locked staging rows with ONE stream synchronisation at the end (two pageable copies were two synchronous transfers
        of their own: 69 -> 45 us per call, which is what a reference-style rollout worker pays per env step)."""
        if eps is not None:
            return self.get_actions(np.asarray(o, np.float32).reshape(1, -1), deterministic, eps)[0].cpu().numpy()
        st = getattr(self, "_stage_act", None)
        if st is None:
            # the policy launches read the observation out of a page-locked row and write the action into another (their device-side
            # addresses, ddrl_host_device_pointer): no copy launches either way — 45 -> ~30 us per call
            hi, ho = torch.empty(1, self.cfg.obs_dim, dtype=torch.float32).pin_memory(), torch.empty(1, self.cfg.act_dim, dtype=torch.float32).pin_memory()
            pi, po = ctypes.c_void_p(), ctypes.c_void_p()
            _lib.check(self._lib.ddrl_host_device_pointer(ctypes.c_void_p(hi.data_ptr()), ctypes.byref(pi)))
            _lib.check(self._lib.ddrl_host_device_pointer(ctypes.c_void_p(ho.data_ptr()), ctypes.byref(po)))
            st = self._stage_act = (hi, hi.numpy(), pi, ho, ho.numpy(), po)
        hi, hiv, pi, ho, hov, po = st
        hiv[0, :] = np.asarray(o, np.float32).reshape(-1)
        a = self.cfg.act_dim
        if getattr(self, "_act_one", True):
            # ONE launch for the one row (ddrl_actor_act_one: both layers, the head, the squash, the noise elements from the counter — the
            # same stream positions _normal(a) would consume); the batched kernels behind a noise launch are a chain of four
            rc = self._lib.ddrl_actor_act_one(self._h, pi, self._noise_seed, self._noise_ctr, 1 if deterministic else 0, po, _lib.stream_ptr())
            if rc == _lib.DDRL_ERR_UNSUPPORTED:
                self._act_one = False
            else:
                _lib.check(rc)
                if not deterministic:
                    self._noise_ctr += a
                torch.cuda.current_stream().synchronize()
                return hov[0].copy()
        e = None if deterministic else self._normal(a)
        _lib.check(self._lib.ddrl_actor_act(self._h, pi, _lib.dptr(e), 1, 1 if deterministic else 0, po, _lib.stream_ptr()))
        torch.cuda.current_stream().synchronize()
        return hov[0].copy()

    def test(self, test_env, replay_buffer=None, n=25):
        """Deterministic evaluation episodes (actor_learner.py:199-218); returns the mean return.  With opt.summary_dir set the
        reference's TensorBoard scalar goes out too: "Reward" = sum(rew) / 25 at step sample_times (actor_learner.py:210-216;
        the divisor is the reference's constant, whatever n is)."""
        rew = []
        for _ in range(n):
            o, r, d, ep_ret, ep_len = test_env.reset(), 0, False, 0, 0
            while not (d or (ep_len == self.opt.max_ep_len)):
                o, r, d, _ = test_env.step(self.get_action(o, True))
                ep_ret += r
                ep_len += 1
            rew.append(ep_ret)
        logdir = getattr(self.opt, "summary_dir", None)
        if logdir and self.job == "main":   # the reference creates its FileWriter for job == "main" only (actor_learner.py:176-179)
            if getattr(self, "_writer", None) is None:
                import datetime
                from .logx import SummaryWriter
                # one run directory per writer, named like the reference's: successive runs do not merge into one curve
                run = "%s-%s-workers_num:%s%%%s" % (datetime.datetime.now(), getattr(self.opt, "env_name", ""), getattr(self.opt, "num_workers", 1),
                                                      getattr(self.opt, "a_l_ratio", ""))
                self._writer = SummaryWriter(logdir + "/" + run)
            sample_times = 0
            if replay_buffer is not None:
                from .workers import _get, _remote
                sample_times = _get(_remote(replay_buffer.get_counts))[0]
            self._writer.add_scalar("Reward", sum(rew) / 25, sample_times)
            self._writer.flush()
        return sum(rew) / n
