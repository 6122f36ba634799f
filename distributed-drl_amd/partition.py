"""Rank-role partitioning of one node (BASELINE.json configs 3 and 4), one process per GPU.

What the reference does with Ray actors:
  * `opt.num_buffers` replay shards; a rollout worker stores into a shard (algos/sac1/sac_ray.py:246), the
    learner draws every batch from ONE shard picked as `np.random.choice(opt.num_buffers, 1)[0]`
    (sac_ray.py:137-141; algos/dqn/train.py:191-199) — a batch never mixes shards;
  * `ps.push` every `push_freq` updates, `ps.pull` by the rollout workers (algos/sac1/sac1.py:149;
    example/dsac.py:59-65); per-node parameter servers / buffers in algos/dqn/train.py:392-411,458.

Here:
  config 3 (2 ranks)  learner on rank 0; BOTH ranks run envs and own a shard (local store, no collective on
                      store); per update the learner picks a shard on its seeded stream; a remote owner draws the
                      batch from its device ring and sends the packed 20 KB to the learner (point-to-point
                      RCCL send/recv over xGMI, prefetched one update ahead); parameters = one RCCL broadcast.
  config 4 (8 ranks)  ranks 0-1 learners (synchronous data parallel: one all-reduce of the flat gradient per
                      update between them — a documented NEW semantics, the reference's multi-learner is
                      unsynchronised last-writer-wins, example/dsac.py:59-62,233), ranks 2-7 rollout ranks with
                      8192 envs and a shard each; rank 0 broadcasts the parameters.
  other sizes         world // 4 learners (at least one); world 1 = config 2 (everything on the one rank).

Every rank derives the whole schedule (which learner draws from which shard at which update) from seeded
streams it holds itself, so no request message is needed: an owner simply issues its sends in schedule order and
a learner its receives — pairwise message order is the match.  Works on any torch.distributed backend; with
"gloo" (functional checks with several ranks on ONE GPU, DDRL_DIST_BACKEND=gloo) device tensors are staged through
the host.
"""
import numpy as np
import torch
import torch.distributed as dist

from . import comm


class Roles:
    """Who learns, who rolls out, who owns which shard."""

    def __init__(self, world, rank=0, num_learners=None):
        self.world, self.rank = int(world), int(rank)
        if self.world == 1:
            self.learners, self.rollouts = [0], [0]
        elif self.world == 2 and num_learners in (None, 1):
            self.learners, self.rollouts = [0], [0, 1]            # config 3: rank 0 learns AND rolls out
        else:
            n_l = max(1, self.world // 4) if num_learners is None else int(num_learners)
            self.learners = list(range(n_l))
            self.rollouts = list(range(n_l, self.world))          # config 4 at world 8: 2 learners + 6 rollout ranks
        self.shard_owner = list(self.rollouts)                     # shard s lives on rank shard_owner[s]

    @property
    def is_learner(self):
        return self.rank in self.learners

    @property
    def is_rollout(self):
        return self.rank in self.rollouts

    @property
    def my_shard(self):
        return self.shard_owner.index(self.rank) if self.rank in self.shard_owner else None

    def describe(self):
        if self.world == 1:
            return "single GPU: envs + replay + learner"
        return "%d learner rank(s) %s%s, %d rollout rank(s) %s with one replay shard each; batch = owner gather + P2P send, params = RCCL broadcast" % (
            len(self.learners), self.learners, " (gradient all-reduce)" if len(self.learners) > 1 else "",
            len(self.rollouts), self.rollouts)


def _is_gloo():
    return dist.is_initialized() and dist.get_backend() == "gloo"


def _send(t, dst):
    if _is_gloo() and t.is_cuda:
        return dist.isend(t.cpu(), dst=dst)
    return dist.isend(t, dst=dst)


class _Recv:
    """A posted receive into `buf` (device); wait() makes the data visible to the current stream."""

    def __init__(self, buf, src):
        self.buf = buf
        if _is_gloo() and buf.is_cuda:
            self.stage = torch.empty(buf.shape, dtype=buf.dtype, device="cpu").pin_memory() if torch.cuda.is_available() else torch.empty(buf.shape, dtype=buf.dtype)
            self.work = dist.irecv(self.stage, src=src)
        else:
            self.stage = None
            self.work = dist.irecv(buf, src=src)

    def wait(self):
        self.work.wait()
        if self.stage is not None:
            self.buf.copy_(self.stage, non_blocking=False)
        return self.buf


def batch_floats(obs_dim, act_dim, B):
    return int(B) * (2 * int(obs_dim) + int(act_dim) + 2)


def batch_views(flat, obs_dim, act_dim, B):
    """The five tensors of a batch as views of one packed float32 buffer [obs1 | obs2 | acts | rews | done]."""
    o, a, B = int(obs_dim), int(act_dim), int(B)
    off, out = 0, {}
    for k, n, shape in (("obs1", B * o, (B, o)), ("obs2", B * o, (B, o)), ("acts", B * a, (B, a)), ("rews", B, (B,)), ("done", B, (B,))):
        out[k] = flat[off:off + n].view(shape)
        off += n
    return out


def sample_packed(rb, B, flat):
    """rb.sample_batch(B) gathered straight into the packed buffer (ONE sampler launch, no copies)."""
    from . import _lib
    v = batch_views(flat, rb.obs_dim, rb.act_dim, B)
    _lib.check(rb._lib.ddrl_replay_sample(rb._h, int(B), _lib.dptr(v["obs1"]), _lib.dptr(v["obs2"]), _lib.dptr(v["acts"]), _lib.dptr(v["rews"]),
                                          _lib.dptr(v["done"]), None, _lib.stream_ptr()))
    return v


class Schedule:
    """The shard every learner draws from at every update: learner l's stream is RandomState(seed + 7919 l), advanced
    once per update like the reference's np.random.choice(num_buffers, 1)[0] (the same MT19937 consumption as
    randint(0, num_buffers), SURVEY §7.2 probe).  Identical on all ranks."""

    def __init__(self, roles, seed=0):
        self.roles = roles
        self.pickers = [comm.ShardPicker(len(roles.shard_owner), seed=int(seed) + 7919 * i) for i in range(len(roles.learners))]

    def next(self):
        """-> [(learner rank, owner rank)] of one update, learner order."""
        return [(l, self.roles.shard_owner[p.next()]) for l, p in zip(self.roles.learners, self.pickers)]


class PartitionedRun:
    """One rank's share of the partitioned actor-learner loop.

    step(n_updates): a rollout rank steps its envs once and stores locally; then, update by update, shard owners
    draw + send the batches the schedule assigns to them and learner ranks receive (one update ahead), train, and —
    with several learners — all-reduce the flat gradient; every `push_freq` updates learner 0 broadcasts the flat
    parameter vector and the rollout ranks adopt it."""

    def __init__(self, opt, roles, make_replay, make_rollout, make_learner, seed=0, push_freq=None, device=None):
        self.opt, self.roles = opt, roles
        self.device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.B = int(opt.batch_size)
        self.nf = batch_floats(opt.obs_dim, opt.act_dim, self.B)
        self.schedule = Schedule(roles, seed)
        self.push_freq = int(getattr(opt, "push_freq", 300) if push_freq is None else push_freq)
        self.rb = make_replay() if roles.my_shard is not None else None
        self.learner = make_learner() if roles.is_learner else None
        from .agent import param_specs
        self.n_params = int(sum(int(np.prod(sh)) for _, sh in param_specs(opt.obs_dim, opt.act_dim, opt.hidden_sizes[0], opt.hidden_sizes[1],
                                                                          ("pi", "q1", "q2"))))
        self.bcast = comm.ParamBroadcast(self.n_params, self.device, src=roles.learners[0]) if roles.world > 1 else None
        self.roll = make_rollout(self.rb) if roles.is_rollout else None
        self.lgroup = None
        if len(roles.learners) > 1 and dist.is_initialized():
            self.lgroup = dist.new_group(ranks=roles.learners)   # collective: every rank calls it
        self.cnt = 0                                  # updates done (per learner)
        self.send_bufs = [torch.empty(self.nf, dtype=torch.float32, device=self.device) for _ in range(4)] if self.rb is not None else []
        self.recv_bufs = [torch.empty(self.nf, dtype=torch.float32, device=self.device) for _ in range(2)] if self.learner is not None else []
        self.sends = []
        self.stats = {"local_batches": 0, "remote_batches": 0, "sent_batches": 0, "pushes": 0}
        self.last_batch = None
        if self.bcast is not None:
            self._push()                              # initial weights: every rank starts from learner 0's vector

    # -- parameters -----------------------------------------------------------------------------
    def _push(self):
        flat = self.learner.get_weights_flat() if self.roles.rank == self.roles.learners[0] else None
        self.bcast.sync(flat)
        self.stats["pushes"] += 1
        if self.learner is not None and self.roles.rank != self.roles.learners[0] and self.stats["pushes"] == 1:
            # data-parallel learners start from learner 0's vector; afterwards identical gradients + identical optimizer
            # state keep them identical, so later pushes leave them alone
            self.learner.set_weights_flat(self.bcast.buf)
        if self.roll is not None:
            n_pi = self.roll.actor.n_params
            self.roll.actor.set_weights_flat(self.bcast.buf[:n_pi])

    # -- one update's traffic -------------------------------------------------------------------
    def _serve(self, plan):
        """Owner side: draw and send the batches this rank owes for one update."""
        me = self.roles.rank
        for l, owner in plan:
            if owner == me and l != me:
                buf = self.send_bufs[self.stats["sent_batches"] % len(self.send_bufs)]
                if len(self.sends) >= len(self.send_bufs):
                    self.sends.pop(0).wait()          # the buffer about to be reused has left
                sample_packed(self.rb, self.B, buf)
                self.sends.append(_send(buf, l))
                self.stats["sent_batches"] += 1

    def _post(self, plan, slot):
        """Learner side: start receiving (or draw locally) the batch of one update into recv slot `slot`."""
        me = self.roles.rank
        for l, owner in plan:
            if l != me:
                continue
            if owner == me:
                sample_packed(self.rb, self.B, self.recv_bufs[slot])
                self.stats["local_batches"] += 1
                return None
            self.stats["remote_batches"] += 1
            return _Recv(self.recv_bufs[slot], owner)
        return None

    def _train(self, flat):
        batch = batch_views(flat, self.opt.obs_dim, self.opt.act_dim, self.B)
        self.last_batch = batch
        if self.lgroup is None:
            self.learner.train_device(batch)
        else:
            g = self.learner.compute_gradients_device(batch)
            comm.allreduce_mean_(g, group=self.lgroup)
            self.learner.apply_gradients(g)

    def step(self, n_updates):
        if self.roll is not None:
            self.roll.step()
        n = int(n_updates)
        if self.roles.world == 1:
            # config 2: everything local — the graph-captured loop is the faster way to run this (workers.TrainDevice)
            for _ in range(n):
                sample_packed(self.rb, self.B, self.recv_bufs[0])
                self._train(self.recv_bufs[0])
                self.cnt += 1
            return
        plans = [self.schedule.next() for _ in range(n)]
        pending = self._post(plans[0], 0) if self.learner is not None else None
        for u in range(n):
            if self.rb is not None:
                self._serve(plans[u])
            if self.learner is not None:
                nxt = self._post(plans[u + 1], (u + 1) & 1) if u + 1 < n else None   # one update ahead
                if pending is not None:
                    pending.wait()
                self._train(self.recv_bufs[u & 1])
                pending = nxt
            self.cnt += 1
            if self.cnt % self.push_freq == 0:
                self._push()
        for w in self.sends:
            w.wait()
        self.sends = []
