"""ReplayBuffer with the reference's surface, backed by the HBM ring of libddrl_hip.so.

Mirrors (same names, argument meaning, error behaviour):
  example/dsac.py:14-48        class ReplayBuffer            -> ReplayBuffer
  algos/sac1/sac1.py:28-63     class ReplayBuffer (counters) -> ReplayBufferSAC1
  algos/dqn/train.py:37-108    class ReplayBuffer(opt, idx)  -> ReplayBufferDQN (+ save/load .npy)

`store` / `sample_batch` keep the reference's host-side types (NumPy in, dict of fresh float32
NumPy arrays out).  The vectorised device path adds `store_batch` / `sample_batch_device`
(torch CUDA tensors in/out, no host round trip); both run the same kernels.
"""
import ctypes
import os

import numpy as np
import torch

from . import _lib


class ReplayBuffer:
    """A simple FIFO experience replay buffer for SAC agents (example/dsac.py:14-48)."""

    _default_batch = 32
    _acts_1d = False

    def __init__(self, obs_dim, act_dim, size, device=None, seed=None, compact_obs=False):
        """compact_obs=True (opt-in): obs1_buf / obs2_buf live as uint8 behind the same float32 surface — for integer-valued pixel
        observations (config 5: 4 M transitions of 84x84x4 frames = 226 GB instead of 903 GB).  Storing anything that is not an
        integer in [0, 255] raises ValueError at the next get_counts / sample_batch / check()."""
        _lib.require_gpu()
        self._lib = _lib.load()
        self.obs_dim, self.act_dim, self.max_size = int(obs_dim), int(act_dim), int(size)
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        self.compact_obs = bool(compact_obs)
        h = ctypes.c_void_p()
        flags = (_lib.DDRL_REPLAY_ACTS_1D if self._acts_1d else 0) | (_lib.DDRL_REPLAY_U8_OBS if self.compact_obs else 0)
        _lib.check(self._lib.ddrl_replay_create(ctypes.byref(h), self.device.index, self.max_size, self.obs_dim,
                                                self.act_dim, flags))
        self._h = h
        self._out = {}
        if seed is not None:
            self.seed(seed)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.ddrl_replay_destroy(h)

    # -- np.random.seed(s) of the reference's buffer process ---------------------------------
    def seed(self, s):
        self._pf_order_store(True)
        _lib.check(self._lib.ddrl_replay_seed(self._h, int(s) & 0xFFFFFFFF, _lib.stream_ptr()))
        self._pf_order_store(False)

    # -- reference surface ---------------------------------------------------------------------
    def store(self, obs, act, rew, next_obs, done, worker_index=None):
        """One transition (example/dsac.py:29-37).  Values are cast to float32 exactly as the
        NumPy row assignment does (float64 -> f32 round-to-nearest, bool -> 0.0/1.0)."""
        # the store launch reads the five values straight out of a page-locked staging row (its device-side address: no copy up, no
        # wait) — one of a ring of rows, each free again once the launch that read it is done (an event per row, looked at when the
        # ring comes round)
        st = getattr(self, "_stage1", None)
        o, a = self.obs_dim, self.act_dim
        if st is None:
            w, R = (2 * o + a + 2 + 15) // 16 * 16, 64
            host = torch.empty(R * w, dtype=torch.float32).pin_memory()
            dp = ctypes.c_void_p()
            _lib.check(self._lib.ddrl_host_device_pointer(ctypes.c_void_p(host.data_ptr()), ctypes.byref(dp)))
            rows = []
            for r in range(R):
                base = dp.value + 4 * r * w
                rows.append((host.numpy()[r * w:(r + 1) * w], [ctypes.c_void_p(base + 4 * k) for k in (0, o, o + a, o + a + 1, 2 * o + a + 1)]))
            st = self._stage1 = {"host": host, "rows": rows, "ev": [None] * R, "turn": 0}
        t = st["turn"]
        st["turn"] = (t + 1) % len(st["rows"])
        if st["ev"][t] is not None:
            st["ev"][t].synchronize()
        else:
            st["ev"][t] = torch.cuda.Event()
        hv, p = st["rows"][t]
        hv[0:o] = np.asarray(obs, dtype=np.float32).reshape(o)
        hv[o:o + a] = np.asarray(act, dtype=np.float32).reshape(a)
        hv[o + a] = np.asarray(rew, dtype=np.float32).reshape(())
        hv[o + a + 1:2 * o + a + 1] = np.asarray(next_obs, dtype=np.float32).reshape(o)
        hv[2 * o + a + 1] = np.asarray(done, dtype=np.float32).reshape(())
        self._pf_order_store(True)
        _lib.check(self._lib.ddrl_replay_store(self._h, p[0], p[1], p[2], p[3], p[4], 1, _lib.stream_ptr()))   # (obs, act, rew, next_obs, done)
        self._pf_order_store(False)
        st["ev"][t].record()

    def store_batch(self, obs, act, rew, next_obs, done):
        """n sequential store() calls in row order (device tensors, float32)."""
        n = int(rew.shape[0])
        obs, act, rew, next_obs, done = (self._f32(t) for t in (obs, act, rew, next_obs, done))
        assert obs.numel() == n * self.obs_dim and next_obs.numel() == n * self.obs_dim
        assert act.numel() == n * self.act_dim and done.numel() == n
        self._pf_order_store(True)
        _lib.check(self._lib.ddrl_replay_store(self._h, _lib.dptr(obs), _lib.dptr(act), _lib.dptr(rew),
                                               _lib.dptr(next_obs), _lib.dptr(done), n, _lib.stream_ptr()))
        self._pf_order_store(False)

    def prefetch(self, batch_size=None, depth=12, hold=2, own_stream=False):
        """The reference's `Cache` (algos/sac1/sac1.py:103-130: a helper that keeps ten sampled batches waiting so that the learner never
        waits for a sample) INSIDE the buffer: from now on `depth - hold` sample_batch(batch_size) draws are always in flight — index
        draw and gather straight into page-locked host blocks, queued on the buffer's stream — and sample_batch(batch_size)
        hands out the oldest one, waiting only if it has not landed yet.  Batches, and their order, are exactly those of the calls
        without prefetch (one sampler, one queue); what changes is WHEN a batch was drawn: up to `depth - hold` calls earlier, so
        transitions stored in between are not in it — the reference's Cache has the same staleness.
        The arrays of a returned batch are views of a ring of host blocks: valid until `hold` further sample_batch calls (copy them to
        keep them longer).  prefetch(0) turns it off.
        own_stream=True: the draws run on a stream of the buffer's own, beside the caller's stream (a learner's update on the same
        thread then no longer queues behind the gather); this object's store / store_batch calls are ordered
        against the draws by events, and so are its other calls that move the sampler or the rows (seed, draws of another size,
        sample_batch_device / sample_many, set_rows); the counters get_counts() reads include the draws in flight.  Writers that go
        to the ring BEHIND this object's back on another stream (a RolloutDevice's fused env-step launch stores through the C
        handle; a feed plan attached with set_feed) are not: keep the default there."""
        pf = getattr(self, "_pf", None)
        if pf is not None:
            if pf["stream"] is not None:
                pf["stream"].synchronize()
            torch.cuda.current_stream().synchronize()
            self._pf = None
        B = int(self._default_batch if batch_size is None else batch_size)
        if B <= 0 or depth <= 0:
            return
        assert 1 <= hold < depth
        o, a = self.obs_dim, self.act_dim
        offs = [0]
        for n in (B * o, B * o, B * a, B, B):
            offs.append((offs[-1] + n + 3) & ~3)
        host = [torch.empty(offs[5], dtype=torch.float32).pin_memory() for _ in range(depth)]
        views = []
        for h in host:
            v = h.numpy()
            views.append(dict(obs1=v[offs[0]:offs[0] + B * o].reshape(B, o), obs2=v[offs[1]:offs[1] + B * o].reshape(B, o),
                              acts=v[offs[2]:offs[2] + B * a] if self._acts_1d else v[offs[2]:offs[2] + B * a].reshape(B, a),
                              rews=v[offs[3]:offs[3] + B], done=v[offs[4]:offs[4] + B]))
        # the gather writes straight into the page-locked blocks (their device-side addresses: posted PCIe writes, no copy launch
        # behind the gather, no device staging buffers) — rows of megabytes (config 5) keep a device block and a copy down
        direct = B * o < (1 << 18)
        dev = None if direct else [torch.empty(offs[5], dtype=torch.float32, device=self.device) for _ in range(depth)]
        ptrs = []
        for i in range(depth):
            if direct:
                dp = ctypes.c_void_p()
                _lib.check(self._lib.ddrl_host_device_pointer(ctypes.c_void_p(host[i].data_ptr()), ctypes.byref(dp)))
                base = dp.value
            else:
                base = dev[i].data_ptr()
            ptrs.append([ctypes.c_void_p(base + 4 * offs[j]) for j in range(5)])
        side = torch.cuda.Stream() if own_stream else None
        if side is not None:
            side.wait_stream(torch.cuda.current_stream())    # what the caller has stored so far is in the ring
        self._pf = dict(B=B, depth=depth, hold=hold, dev=dev, host=host, views=views, ptrs=ptrs, ev=[torch.cuda.Event() for _ in range(depth)], head=0,
                        stream=side, sptr=None if side is None else ctypes.c_void_p(side.cuda_stream), store_ev=None, draw_ev=None)
        for i in range(depth - hold):
            self._pf_enqueue(i)

    def _pf_enqueue(self, i):
        pf = self._pf
        p = pf["ptrs"][i]
        side = pf["stream"]
        if side is None:
            _lib.check(self._lib.ddrl_replay_sample(self._h, pf["B"], p[0], p[1], p[2], p[3], p[4], None, _lib.stream_ptr()))
            if pf["dev"] is not None:
                pf["host"][i].copy_(pf["dev"][i], non_blocking=True)
            pf["ev"][i].record()
            return
        if pf["store_ev"] is not None:
            side.wait_event(pf["store_ev"])                  # the caller's last store has landed
            pf["store_ev"] = None
        _lib.check(self._lib.ddrl_replay_sample(self._h, pf["B"], p[0], p[1], p[2], p[3], p[4], None, pf["sptr"]))
        if pf["dev"] is None:
            pf["ev"][i].record(side)
            pf["draw_ev"] = pf["ev"][i]                      # (a store behind this draw must not overwrite rows it is still gathering)
            return
        pf["draw_ev"] = torch.cuda.Event()
        pf["draw_ev"].record(side)
        with torch.cuda.stream(side):
            pf["host"][i].copy_(pf["dev"][i], non_blocking=True)
        pf["ev"][i].record(side)

    def _pf_order_store(self, before):
        """own-stream prefetch: a store waits for the draws in flight (before=True), and leaves an event the next draw waits for."""
        pf = getattr(self, "_pf", None)
        if pf is None or pf["stream"] is None:
            return
        if before:
            if pf["draw_ev"] is not None:
                torch.cuda.current_stream().wait_event(pf["draw_ev"])
        else:
            pf["store_ev"] = torch.cuda.Event()
            pf["store_ev"].record()

    def sample_batch(self, batch_size=None):
        """dict(obs1, obs2, acts, rews, done) of fresh float32 NumPy arrays (example/dsac.py:39-45).
        Raises ValueError("high <= 0") on an empty buffer like the reference.  One call at a time per buffer object (a Ray actor's calls
        are serial, and so are those of remote.py's actor threads): the batch passes through a page-locked block the object owns."""
        pf = getattr(self, "_pf", None)
        if pf is not None and pf["B"] == int(self._default_batch if batch_size is None else batch_size):
            i, D = pf["head"], pf["depth"]
            pf["head"] = (i + 1) % D
            pf["ev"][i].synchronize()
            self._pf_enqueue((i + D - pf["hold"]) % D)       # the block handed out `hold` calls ago goes back into flight
            return pf["views"][i]
        # gathered into ONE packed device block and brought down with one copy (five device-to-host copies were most of the call);
        # the five arrays are disjoint pieces of that fresh host block
        B = int(self._default_batch if batch_size is None else batch_size)
        o, a = self.obs_dim, self.act_dim
        if B * o >= (1 << 22):    # rows of megabytes (config 5): the gather's own aligned per-array buffers
            d = self.sample_batch_device(B, fresh=True)
            return {k: v.cpu().numpy() for k, v in d.items()}
        offs = [0]
        for n in (B * o, B * o, B * a, B, B):
            offs.append((offs[-1] + n + 3) & ~3)               # every piece 16-byte aligned
        # ... gathered straight into a page-locked block of this buffer (its device-side address: no device staging block, no copy
        # launch), of which the caller gets a fresh copy: 38 -> ~20 us per call
        st = self.__dict__.setdefault("_plain_host", {}).get(B)
        if st is None:
            host = torch.empty(offs[5], dtype=torch.float32).pin_memory()
            dp = ctypes.c_void_p()
            _lib.check(self._lib.ddrl_host_device_pointer(ctypes.c_void_p(host.data_ptr()), ctypes.byref(dp)))
            st = self._plain_host[B] = (host, host.numpy(), [ctypes.c_void_p(dp.value + 4 * offs[j]) for j in range(5)])
        p = st[2]
        self._pf_order_store(True)       # (a draw of another size while a prefetch is on: behind the draws in flight, in front of the next)
        _lib.check(self._lib.ddrl_replay_sample(self._h, B, p[0], p[1], p[2], p[3], p[4], None, _lib.stream_ptr()))
        self._pf_order_store(False)
        torch.cuda.current_stream().synchronize()
        h = st[1].copy()
        return dict(obs1=h[offs[0]:offs[0] + B * o].reshape(B, o), obs2=h[offs[1]:offs[1] + B * o].reshape(B, o),
                    acts=h[offs[2]:offs[2] + B * a] if self._acts_1d else h[offs[2]:offs[2] + B * a].reshape(B, a),
                    rews=h[offs[3]:offs[3] + B], done=h[offs[4]:offs[4] + B])

    def sample_batch_device(self, batch_size=None, fresh=False, with_indices=False):
        B = int(self._default_batch if batch_size is None else batch_size)
        out = self._buffers(B, fresh)
        idx = None
        if with_indices:
            idx = torch.empty(B, dtype=torch.int64, device=self.device)
        self._pf_order_store(True)
        _lib.check(self._lib.ddrl_replay_sample(self._h, B, _lib.dptr(out["obs1"]), _lib.dptr(out["obs2"]),
                                                _lib.dptr(out["acts"]), _lib.dptr(out["rews"]),
                                                _lib.dptr(out["done"]), _lib.dptr(idx), _lib.stream_ptr()))
        self._pf_order_store(False)
        if with_indices:
            out = dict(out, idxs=idx)
        return out

    def gather_device(self, idx, fresh=True):
        """The five fancy-index gathers for caller-supplied int64 device indices."""
        B = int(idx.numel())
        out = self._buffers(B, fresh)
        idx = idx.to(device=self.device, dtype=torch.int64).contiguous()
        _lib.check(self._lib.ddrl_replay_gather(self._h, _lib.dptr(idx), B, _lib.dptr(out["obs1"]),
                                                _lib.dptr(out["obs2"]), _lib.dptr(out["acts"]),
                                                _lib.dptr(out["rews"]), _lib.dptr(out["done"]), _lib.stream_ptr()))
        return out

    def sample_many(self, batch_size, count, flat):
        """`count` consecutive sample_batch(batch_size) draws gathered into the packed float32 device buffer `flat`
        as one block [obs1 | obs2 | acts | rews | done], each array [count * B, w] (ddrl_replay_sample_many): what a
        shard owner sends a remote learner for one step (partition.py)."""
        B, K = int(batch_size), int(count)
        ptrs, off = (ctypes.c_void_p * 5)(), 0
        for j, w in enumerate((self.obs_dim, self.obs_dim, self.act_dim, 1, 1)):
            ptrs[j] = flat.data_ptr() + 4 * off
            off += K * B * w
        assert flat.numel() >= off and flat.dtype == torch.float32 and flat.is_cuda and flat.is_contiguous()
        self._pf_order_store(True)
        _lib.check(self._lib.ddrl_replay_sample_many(self._h, B, K, ptrs, _lib.stream_ptr()))
        self._pf_order_store(False)
        return flat[:off]

    def set_feed(self, plan, batch_size, regions):
        """Attach a per-update feed plan (int32 device tensor: -1 = local draw, r << 24 | i = batch i of regions[r]) to
        this ring's sampler; regions = [(packed block tensor, batches in it)].  plan=None detaches."""
        if plan is None:
            _lib.check(self._lib.ddrl_replay_set_feed(self._h, None, 0, 0, 0, None, None, _lib.stream_ptr()))
            self._feed_keep = None
            return
        assert plan.dtype == torch.int32 and plan.is_cuda and plan.is_contiguous()
        n = len(regions)
        base, cnt = (ctypes.c_void_p * max(n, 1))(), (ctypes.c_int32 * max(n, 1))()
        for r, (t, k) in enumerate(regions):
            assert t.is_cuda and t.is_contiguous() and t.dtype == torch.float32
            base[r], cnt[r] = t.data_ptr(), int(k)
        _lib.check(self._lib.ddrl_replay_set_feed(self._h, _lib.dptr(plan), int(plan.numel()), int(batch_size), n, base, cnt, _lib.stream_ptr()))
        self._feed_keep = (plan, [t for t, _ in regions])   # the kernels read these until the next set_feed

    def take_error(self, out):
        """Move the sampler's sticky device-side error into out[0] (int32 device tensor) and clear it; no sync."""
        assert out.dtype == torch.int32 and out.is_cuda
        _lib.check(self._lib.ddrl_replay_take_error(self._h, _lib.dptr(out), _lib.stream_ptr()))

    def get_counts(self):
        """example/dsac.py:47-48: number of store() calls so far."""
        return self._counts()[2]

    # -- helpers -------------------------------------------------------------------------------
    def _f32(self, t):
        if not torch.is_tensor(t):
            t = torch.as_tensor(np.asarray(t, dtype=np.float32))
        return t.to(device=self.device, dtype=torch.float32).contiguous()

    def _buffers(self, B, fresh):
        if not fresh and B in self._out:
            return self._out[B]
        e = lambda *s: torch.empty(*s, dtype=torch.float32, device=self.device)
        out = dict(obs1=e(B, self.obs_dim), obs2=e(B, self.obs_dim),
                   acts=e(B) if self._acts_1d else e(B, self.act_dim), rews=e(B), done=e(B))
        if not fresh:
            self._out[B] = out
        return out

    def _counts(self):
        c = [ctypes.c_int64() for _ in range(4)]
        _lib.check(self._lib.ddrl_replay_counts(self._h, *[ctypes.byref(x) for x in c], _lib.stream_ptr()))
        return tuple(int(x.value) for x in c)  # (ptr, size, steps, sample_times)

    @property
    def ptr(self):
        return self._counts()[0]

    @property
    def size(self):
        return self._counts()[1]

    _RING_NAMES = ("obs1_buf", "obs2_buf", "acts_buf", "rews_buf", "done_buf")

    def _ring_shapes(self):
        N = self.max_size
        return [(N, self.obs_dim), (N, self.obs_dim), (N,) if self._acts_1d else (N, self.act_dim), (N,), (N,)]

    def rings(self):
        """Views of the five rings as torch tensors (device memory owned by the handle).  Compact observation arrays come back
        as float32 COPIES (rows_export) — the float32 values the reference's arrays would hold."""
        p = [ctypes.c_void_p() for _ in range(5)]
        _lib.check(self._lib.ddrl_replay_buffers(self._h, *[ctypes.byref(x) for x in p]))
        out = {}
        for j, (n, ptr, s) in enumerate(zip(self._RING_NAMES, p, self._ring_shapes())):
            out[n] = self.rows(j, 0, self.max_size).reshape(s) if (self.compact_obs and j < 2) else _view(ptr.value, s, self.device)
        return out

    def rows(self, array, row0, nrows):
        """Ring rows [row0, row0 + nrows) of array 0..4 as a fresh float32 device tensor, whatever the storage kind."""
        w = int(np.prod(self._ring_shapes()[array][1:])) if len(self._ring_shapes()[array]) > 1 else 1
        out = torch.empty(int(nrows), w, dtype=torch.float32, device=self.device)
        _lib.check(self._lib.ddrl_replay_rows_export(self._h, int(array), int(row0), int(nrows), _lib.dptr(out), _lib.stream_ptr()))
        return out

    def set_rows(self, array, row0, values):
        values = self._f32(values)
        w = int(np.prod(self._ring_shapes()[array][1:])) if len(self._ring_shapes()[array]) > 1 else 1
        nrows = values.numel() // w
        self._pf_order_store(True)
        _lib.check(self._lib.ddrl_replay_rows_import(self._h, int(array), int(row0), int(nrows), _lib.dptr(values), _lib.stream_ptr()))
        self._pf_order_store(False)

    def check(self):
        """Surface the ring's sticky device-side error now (an empty-ring draw inside a graph, a value a compact array cannot hold)."""
        self._counts()

    def mt_state(self):
        key = np.empty(624, dtype=np.uint32)
        pos = ctypes.c_int32()
        _lib.check(self._lib.ddrl_replay_mt_state(self._h, ctypes.c_void_p(key.ctypes.data), ctypes.byref(pos),
                                                  _lib.stream_ptr()))
        return key, int(pos.value)


class ReplayBufferSAC1(ReplayBuffer):
    """algos/sac1/sac1.py:28-63: default batch 128, get_counts -> (sample_times, steps, size)."""
    _default_batch = 128

    def get_counts(self):
        _, size, steps, samples = self._counts()
        return samples, steps, size


class ReplayBufferDQN(ReplayBuffer):
    """algos/dqn/train.py:37-108: ReplayBuffer(opt, buffer_index), 1-D acts_buf, batch size from
    opt, get_counts -> (learner_steps, actor_steps, size), save/load of the .npy checkpoint."""
    _acts_1d = True

    CHUNK_ROWS = 4096   # rows per device <-> host transfer of save / load (462 MB of float32 at config 5's row width)

    def __init__(self, opt, buffer_index, device=None, seed=None, compact_obs=None):
        """compact_obs (default: opt.compact_obs, else False): uint8 storage of the pixel observations behind the float32 surface."""
        self.opt, self.buffer_index = opt, buffer_index
        compact = bool(getattr(opt, "compact_obs", False)) if compact_obs is None else bool(compact_obs)
        super().__init__(opt.obs_dim, 1, opt.buffer_size, device=device, seed=seed, compact_obs=compact)

    def store(self, obs, act, rew, next_obs, done, worker_index=None):
        super().store(obs, act, rew, next_obs, done)

    def sample_batch(self, batch_size=None):
        return super().sample_batch(self.opt.batch_size if batch_size is None else batch_size)

    def get_counts(self):
        _, size, steps, samples = self._counts()
        return samples, steps, size

    def _ckpt(self, checkpoint_path=None):
        return checkpoint_path if checkpoint_path else os.path.join(self.opt.save_dir, "checkpoint")

    def save(self, checkpoint_path=None):
        """Five .npy arrays + buffer_infos (ptr, size, max_size, actor_steps, learner_steps)
        — the on-disk format of algos/dqn/train.py:82-90."""
        path = self._ckpt(checkpoint_path)
        os.makedirs(path, exist_ok=True)
        for j, (name, shape) in enumerate(zip(self._RING_NAMES, self._ring_shapes())):
            # float32 .npy whatever the storage kind, written chunk by chunk (the arrays do not fit host memory at config 5's size)
            mm = np.lib.format.open_memmap(os.path.join(path, "%s-%s.npy" % (name, self.buffer_index)), mode="w+", dtype=np.float32, shape=shape)
            for r0 in range(0, self.max_size, self.CHUNK_ROWS):
                n = min(self.CHUNK_ROWS, self.max_size - r0)
                mm[r0:r0 + n] = self.rows(j, r0, n).cpu().numpy().reshape((n,) + shape[1:])
            mm.flush()
            del mm
        ptr, size, steps, samples = self._counts()
        np.save(os.path.join(path, "buffer_infos-%s" % self.buffer_index),
                np.array((ptr, size, self.max_size, steps, samples)))

    def load(self, checkpoint_path=None):
        """algos/dqn/train.py:92-108."""
        path = self._ckpt(checkpoint_path)
        for j, name in enumerate(self._RING_NAMES):
            arr = np.load(os.path.join(path, "%s-%s.npy" % (name, self.buffer_index)), mmap_mode="r")
            for r0 in range(0, self.max_size, self.CHUNK_ROWS):
                n = min(self.CHUNK_ROWS, self.max_size - r0)
                self.set_rows(j, r0, torch.from_numpy(np.array(arr[r0:r0 + n], dtype=np.float32)))
        infos = np.load(os.path.join(path, "buffer_infos-%s.npy" % self.buffer_index))
        _lib.check(self._lib.ddrl_replay_set_counts(self._h, int(infos[0]), int(infos[1]), int(infos[3]),
                                                    int(infos[4]), _lib.stream_ptr()))


class ReplayBufferNStep:
    """The n-step window buffer of algos/sac1/sac_ray.py:34-82: `ReplayBuffer(opt)` with
    opt.buffer_size slots, each holding (opt.Ln + 1) observation frames and opt.Ln (action, reward,
    done) triples; `store(o_queue, a_r_d_queue, worker_index)` takes the rollout's two deques
    (sac_ray.py:53-70), `sample_batch()` returns dict(obs, acts, rews, done) of whole windows
    (sac_ray.py:72-80) and the counters advance by opt.num_buffers per call.  Float observations
    only (the reference's packed-string CNN frames are out of scope)."""

    def __init__(self, opt, device=None, seed=None):
        _lib.require_gpu()
        self._lib = _lib.load()
        self.opt = opt
        self.Ln = int(opt.Ln)
        self.obs_shape, self.act_shape = tuple(opt.obs_shape), tuple(opt.act_shape)
        self.obs_dim = int(np.prod(self.obs_shape)) if self.obs_shape else 1
        self.act_dim = int(np.prod(self.act_shape)) if self.act_shape else 1
        self.max_size = int(opt.buffer_size)
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        self.widths = [(self.Ln + 1) * self.obs_dim, self.Ln * self.act_dim, self.Ln, self.Ln]
        self.shapes = [(self.Ln + 1,) + self.obs_shape, (self.Ln,) + self.act_shape, (self.Ln,), (self.Ln,)]
        self.names = ["obs", "acts", "rews", "done"]
        h = ctypes.c_void_p()
        w = (ctypes.c_int32 * 4)(*self.widths)
        nb = int(getattr(opt, "num_buffers", 1))
        _lib.check(self._lib.ddrl_replay_create_ex(ctypes.byref(h), self.device.index, self.max_size, 4, w, nb, nb))
        self._h = h
        if seed is not None:
            self.seed(seed)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.ddrl_replay_destroy(h)

    def seed(self, s):
        _lib.check(self._lib.ddrl_replay_seed(self._h, int(s) & 0xFFFFFFFF, _lib.stream_ptr()))

    def store(self, o_queue, a_r_d_queue, worker_index=None):
        """One window (sac_ray.py:53-70): o_queue = Ln+1 tuples (o,), a_r_d_queue = Ln tuples (a, r, d)."""
        obs = np.stack([np.asarray(q[0], dtype=np.float32) for q in o_queue], axis=0)
        a = np.stack([np.asarray(q[0], dtype=np.float32) for q in a_r_d_queue], axis=0)
        r = np.array([q[1] for q in a_r_d_queue], dtype=np.float32)
        d = np.array([q[2] for q in a_r_d_queue], dtype=np.float32)
        self.store_batch(*(torch.from_numpy(np.ascontiguousarray(x)).to(self.device).reshape(1, -1) for x in (obs, a, r, d)))

    def store_batch(self, obs, acts, rews, done):
        """n windows at once (device tensors [n, ...]) == n sequential store() calls."""
        ts = [t.to(device=self.device, dtype=torch.float32).contiguous() for t in (obs, acts, rews, done)]
        n = int(ts[2].shape[0])
        for t, w in zip(ts, self.widths):
            assert t.numel() == n * w
        ptrs = (ctypes.c_void_p * 4)(*[t.data_ptr() for t in ts])
        _lib.check(self._lib.ddrl_replay_store_ex(self._h, ptrs, n, _lib.stream_ptr()))

    def store_masked(self, obs, acts, rews, done, mask):
        """store() for the rows with mask != 0, in row order (device uint8[n]); the row count stays on the
        device — get_counts() refreshes the host view."""
        ts = [t if (t.dtype == torch.float32 and t.is_contiguous()) else t.to(dtype=torch.float32).contiguous() for t in (obs, acts, rews, done)]
        n = int(ts[2].shape[0])
        for t, w in zip(ts, self.widths):
            assert t.numel() == n * w
        m = mask if (mask.dtype == torch.uint8 and mask.is_contiguous()) else mask.to(dtype=torch.uint8).contiguous()
        ptrs = (ctypes.c_void_p * 4)(*[t.data_ptr() for t in ts])
        _lib.check(self._lib.ddrl_replay_store_masked_ex(self._h, ptrs, _lib.dptr(m), n, _lib.stream_ptr()))

    def sample_batch_device(self, batch_size=None, with_indices=False):
        B = int(self.opt.batch_size if batch_size is None else batch_size)
        outs = [torch.empty((B,) + s, dtype=torch.float32, device=self.device) for s in self.shapes]
        ptrs = (ctypes.c_void_p * 4)(*[t.data_ptr() for t in outs])
        idx = torch.empty(B, dtype=torch.int64, device=self.device) if with_indices else None
        _lib.check(self._lib.ddrl_replay_sample_ex(self._h, B, ptrs, _lib.dptr(idx), _lib.stream_ptr()))
        d = dict(zip(self.names, outs))
        if with_indices:
            d["idxs"] = idx
        return d

    def sample_batch(self):
        return {k: v.cpu().numpy() for k, v in self.sample_batch_device().items()}

    def get_counts(self):
        c = [ctypes.c_int64() for _ in range(4)]
        _lib.check(self._lib.ddrl_replay_counts(self._h, *[ctypes.byref(x) for x in c], _lib.stream_ptr()))
        ptr, size, steps, samples = (int(x.value) for x in c)
        return samples, steps, size

    def rings(self):
        p = (ctypes.c_void_p * 6)()
        _lib.check(self._lib.ddrl_replay_buffers_ex(self._h, p, None, None))
        return {"buffer_" + n[0]: _view(p[j], (self.max_size,) + s, self.device)
                for j, (n, s) in enumerate(zip(self.names, self.shapes))}

    def mt_state(self):
        key = np.empty(624, dtype=np.uint32)
        pos = ctypes.c_int32()
        _lib.check(self._lib.ddrl_replay_mt_state(self._h, ctypes.c_void_p(key.ctypes.data), ctypes.byref(pos),
                                                  _lib.stream_ptr()))
        return key, int(pos.value)


def _view(ptr, shape, device):
    """Wrap raw device memory owned by a handle as a float32 torch tensor (no copy)."""
    n = int(np.prod(shape))

    class _Holder:
        __cuda_array_interface__ = {"shape": (n,), "typestr": "<f4", "data": (int(ptr), False), "version": 2}
    return torch.as_tensor(_Holder(), device=device).view(*shape)
