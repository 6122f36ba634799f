"""Rank-role partitioning of one node (BASELINE.json configs 3 and 4), one process per GPU.

What the reference does with Ray actors:
  * `opt.num_buffers` replay shards; a rollout worker stores into a shard (algos/sac1/sac_ray.py:246), the
    learner draws every batch from ONE shard picked as `np.random.choice(opt.num_buffers, 1)[0]`
    (sac_ray.py:137-141; algos/dqn/train.py:191-199) — a batch never mixes shards;
  * `ps.push` every `push_freq` updates, `ps.pull` by the rollout workers (algos/sac1/sac1.py:149;
    example/dsac.py:59-65); per-node parameter servers / buffers in algos/dqn/train.py:392-411,458.

Here:
  config 3 (2 ranks)  learner on rank 0; BOTH ranks run envs and own a shard (local store, no collective on
                      store); per update the learner picks a shard on its seeded stream; the batches a remote owner
                      owes for one step are drawn from its device ring in one launch sequence
                      (ddrl_replay_sample_many) and sent as ONE point-to-point message (RCCL send/recv over xGMI,
                      20 KB per batch); the learner's sampler follows the step's plan from device memory (local
                      draw or "batch i of owner r's block", ddrl_replay_set_feed), so the graph-captured learner
                      loop of config 2 runs unchanged; parameters = one RCCL broadcast.
  config 4 (8 ranks)  ranks 0-1 learners (synchronous data parallel: one all-reduce of the flat gradient per
                      update between them — a documented NEW semantics, the reference's multi-learner is
                      unsynchronised last-writer-wins, example/dsac.py:59-62,233), ranks 2-7 rollout ranks with
                      8192 envs and a shard each; rank 0 broadcasts the parameters.
  other sizes         world // 4 learners (at least one); world 1 = config 2 (everything on the one rank).

Every rank derives the whole schedule (which learner draws from which shard at which update) from seeded
streams it holds itself, so no request message is needed: per step an owner sends each learner one block and a
learner posts one receive per owner.  The rings do not change between a step's env step and its updates, so
drawing a step's batches ahead of the updates hands out exactly the batches per-update requests would; a shard
serving two learners serves learner 0's batches of the step first (the reference's order is whatever order the
Ray calls arrive in).  Works on any torch.distributed backend; with
"gloo" (functional checks with several ranks on ONE GPU, DDRL_DIST_BACKEND=gloo) device tensors are staged through
the host.
"""
import numpy as np
import torch
import torch.distributed as dist

from . import comm


MAX_FEED_REGIONS = 16     # csrc/replay_device.h: MAX_FEED


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
        if not self.learners or not self.rollouts:
            raise ValueError("world %d with %d learner rank(s) leaves no rollout rank / replay shard: need at least one of each"
                             % (self.world, len(self.learners)))
        if len(self.shard_owner) > MAX_FEED_REGIONS:
            raise ValueError("%d shard owners: a learner's sampler follows at most %d remote blocks per step (ddrl_replay_set_feed)"
                             % (len(self.shard_owner), MAX_FEED_REGIONS))

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
        return "%d learner rank(s) %s%s, %d rollout rank(s) %s with one replay shard each; batches = owner gather, one P2P block per (owner, learner) and step, params = RCCL broadcast" % (
            len(self.learners), self.learners, " (gradient all-reduce)" if len(self.learners) > 1 else "",
            len(self.rollouts), self.rollouts)


PLAN_STAGES = 4           # pinned staging buffers for the step plan = how many steps the host may run ahead


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
            # pinned staging: the host allocator keeps the block alive until the copy's event has passed
            self.buf.copy_(self.stage, non_blocking=self.stage.is_pinned())
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


class _Loop:
    """ddrl_loop over (learner, ring): n updates of sample -> train per call, graph-captured in chunks."""

    def __init__(self, learner, rb, per_graph):
        import ctypes
        from . import _lib
        self._lib, self._libmod = _lib.load(), _lib
        h = ctypes.c_void_p()
        _lib.check(self._lib.ddrl_loop_create(ctypes.byref(h), learner._h, rb._h, int(per_graph), int(learner._noise_seed)))
        self._h, self._keep = h, (learner, rb)

    def run(self, n):
        self._libmod.check(self._lib.ddrl_loop_run(self._h, int(n), self._libmod.stream_ptr()))

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.ddrl_loop_destroy(h)


class PartitionedRun:
    """One rank's share of the partitioned actor-learner loop.

    step(n_updates): a rollout rank steps its envs once and stores locally; shard owners draw the blocks of batches
    the schedule assigns to them for this step and send one message per learner; a learner rank receives its
    blocks, hands the step's plan to its sampler and runs the n updates — the graph-captured loop when it is the
    only learner, eager compute / all-reduce / apply per update with several; every `push_freq` updates learner 0
    broadcasts the flat parameter vector and the rollout ranks adopt it."""

    def __init__(self, opt, roles, make_replay, make_rollout, make_learner, seed=0, push_freq=None, device=None, updates_per_graph=16,
                 force_dp=False, dp_updates_per_graph=None, free_steps=0):
        """free_steps = K > 0: the FREE-RUNNING mode (example/dsac.py:229-236: rollouts and learners are launched and left running, no
        gate between them, dsac.py:76-150).  A step is then K vector steps on every rollout rank beside the learners' n updates, and
        nothing on a rollout rank's env stream ever waits for the learners of the SAME step:
          * blocks: an owner draws and sends, behind its K vector steps of step s, the batches of step s + 1; a learner posts the
            receives for step s + 1 BEFORE it runs the updates of step s (lock-step mode waits for a step's blocks in front of
            its updates) — the transfer runs under the updates;
          * pushes: a rollout rank joins the step's broadcasts on a communication stream of its own and adopts the last of them
            at the start of its NEXT step (per env at episode ends, as always) — its env stream waits for learner step s - 1,
            never for learner step s.
        The batches of step s + 1 are therefore drawn from the rings as they stand after the env steps of step s, one step
        earlier than in lock-step mode: which transitions an update sees differs (the reference's own timing is not defined at
        all), every index stream and the shard schedule stay exactly the same.  0 (default): lock-step, what the provenance
        tests pin."""
        self.opt, self.roles = opt, roles
        self.free_steps = int(free_steps)
        self.pending, self.regions_free = None, {}
        self.comm_stream, self.push_event, self.adopt_event, self.push_pending = None, None, None, False
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
        # (a world-size-1 process group still broadcasts: the one-GPU box drives the same RCCL calls as N > 1)
        self.bcast = comm.ParamBroadcast(self.n_params, self.device, src=roles.learners[0]) if (roles.world > 1 or dist.is_initialized()) else None
        self.roll = make_rollout(self.rb) if roles.is_rollout else None
        self.lgroup = None
        if (len(roles.learners) > 1 or force_dp) and dist.is_initialized():   # force_dp: the data-parallel step with a group of one
            self.lgroup = dist.new_group(ranks=roles.learners)   # collective: every rank calls it
        self.cnt = 0                                  # updates done (per learner)
        self.feed_ring, self.loop, self.batch_buf = None, None, None
        if self.learner is not None:
            # the ring whose sampler the learner drives: its own shard, or (dedicated learner rank) an empty one-row ring
            # that only ever follows the feed plan
            from .replay import ReplayBufferSAC1
            self.feed_ring = self.rb if self.rb is not None else ReplayBufferSAC1(opt.obs_dim, opt.act_dim, 1)
            self.batch_buf = torch.empty(self.nf, dtype=torch.float32, device=self.device)
            if self.lgroup is None:
                self.loop = _Loop(self.learner, self.feed_ring, updates_per_graph)
            else:
                self.dp_grads, self.dp_apply, self.dp_g = self.learner.dp_stepper(self.feed_ring)
                # the data-parallel step captured as ONE graph of k updates (gradients -> RCCL all-reduce -> apply, k times): RCCL
                # collectives are stream-capturable, gloo's (host staging) are not.  0 = eager.  DDRL_DP_GRAPH overrides.
                import os
                k = int(os.environ.get("DDRL_DP_GRAPH", updates_per_graph if dp_updates_per_graph is None else dp_updates_per_graph))
                self.dp_per_graph = (k & ~1) if (dist.is_initialized() and dist.get_backend(self.lgroup) == "nccl") else 0
                self.dp_graph = None
        self.send_bufs, self.regions = {}, {}
        # the step's plan goes host -> device with an async copy while the host runs ahead of the device by whole
        # steps: a pinned staging buffer is rewritten only after the event behind its last copy has completed
        self.plan_d, self.plan_stage, self.plan_turn = None, [], 0
        self.err_d = torch.zeros(1, dtype=torch.int32, device=self.device) if self.learner is not None else None
        self.err_stage = []                           # [(pinned int32[1], event)] of the steps in flight
        # every rank, not only the learners, is held to PLAN_STAGES steps of run-ahead: a rollout / shard-owner rank never blocks
        # its host otherwise (env step, block draw, isend, Work.wait and the broadcast are all stream-ordered under RCCL), and
        # under the actor/learner gate it is ~4000x faster than the learner it serves — it would queue a whole run's sends and
        # broadcasts at once and the ones at the back would sit in the queue past the process group's watchdog timeout
        self.step_events = []
        self.sends = []
        self.stats = {"local_batches": 0, "remote_batches": 0, "sent_batches": 0, "sent_blocks": 0, "pushes": 0}
        self.last_plan = None
        import os
        self.timing = os.environ.get("DDRL_PART_TIMING", "0") == "1"
        if self.bcast is not None:
            self._push()                              # initial weights: every rank starts from learner 0's vector

    # -- parameters -----------------------------------------------------------------------------
    def _push(self):
        if self.free_steps > 0 and self.learner is None and self.roll is not None and self.stats["pushes"] >= 1:
            # free-running rollout rank: the collective runs on a stream of its own, the env stream goes on; the weights are adopted
            # at the start of the next step (step(): _adopt_pushed)
            if self.comm_stream is None:
                self.comm_stream = torch.cuda.Stream()
            with torch.cuda.stream(self.comm_stream):
                if self.adopt_event is not None:
                    self.comm_stream.wait_event(self.adopt_event)   # the pack kernel of the last adoption has read the buffer
                self.bcast.sync(None)
                self.push_event = torch.cuda.Event()
                self.push_event.record()
            self.push_pending = True
            self.stats["pushes"] += 1
            return
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

    def _adopt_pushed(self):
        """Free mode, rollout rank: the last broadcast of the previous step becomes the newest policy version."""
        if not self.push_pending:
            return
        torch.cuda.current_stream().wait_event(self.push_event)
        n_pi = self.roll.actor.n_params
        self.roll.actor.set_weights_flat(self.bcast.buf[:n_pi])
        self.adopt_event = torch.cuda.Event()
        self.adopt_event.record()
        self.push_pending = False

    # -- one step's traffic ---------------------------------------------------------------------
    def _buf(self, table, key, batches):
        t = table.get(key)
        if t is None or t.numel() < batches * self.nf:
            t = table[key] = torch.empty(max(batches, 64) * self.nf, dtype=torch.float32, device=self.device)
        return t

    def _serve(self, plans):
        """Owner side: draw and send, learner by learner, the block of batches this rank owes for the step."""
        me = self.roles.rank
        for li, l in enumerate(self.roles.learners):
            if l == me:
                continue
            k = sum(1 for p in plans if p[li][1] == me)
            if k == 0:
                continue
            blk = self.rb.sample_many(self.B, k, self._buf(self.send_bufs, l, k))
            self.sends.append(_send(blk, l))
            self.stats["sent_batches"] += k
            self.stats["sent_blocks"] += 1

    def _receive(self, plans):
        """Learner side: post one receive per remote owner, lay the step's plan down for the sampler."""
        self._arm(self._post(plans, self.regions))

    def _post(self, plans, table, parity=None):
        """Post the receives of one step's blocks (into `table`'s buffers; free mode: one table per step parity)."""
        me, li = self.roles.rank, self.roles.learners.index(self.roles.rank)
        owners = [p[li][1] for p in plans]
        remote = sorted(set(o for o in owners if o != me))
        count = {o: owners.count(o) for o in remote}
        recvs, regions = [], []
        for o in remote:
            buf = self._buf(table, o if parity is None else (o, parity), count[o])[: count[o] * self.nf]
            recvs.append(_Recv(buf, o))
            regions.append((buf, count[o]))
        return {"owners": owners, "remote": remote, "count": count, "recvs": recvs, "regions": regions}

    def _arm(self, rec):
        """The posted blocks of a step become the sampler's feed: plan to the device, set_feed, wait for the blocks."""
        me = self.roles.rank
        owners, remote, count, recvs, regions = rec["owners"], rec["remote"], rec["count"], rec["recvs"], rec["regions"]
        nxt = {o: 0 for o in remote}
        plan = np.empty(len(owners), dtype=np.int32)
        for u, o in enumerate(owners):
            if o == me:
                plan[u] = -1
            else:
                plan[u] = (remote.index(o) << 24) | nxt[o]
                nxt[o] += 1
        n = len(owners)
        if self.plan_d is None or self.plan_d.numel() < n:
            self.plan_d = torch.empty(max(n, 64), dtype=torch.int32, device=self.device)
            self.plan_stage = [[torch.empty(max(n, 64), dtype=torch.int32).pin_memory(), None] for _ in range(PLAN_STAGES)]
        stage = self.plan_stage[self.plan_turn % PLAN_STAGES]
        self.plan_turn += 1
        if stage[1] is not None:
            stage[1].synchronize()                    # the copy that last read this staging buffer is done
        stage[0][:n].copy_(torch.from_numpy(plan))
        self.plan_d[:n].copy_(stage[0][:n], non_blocking=True)   # stream order puts it behind the previous step's updates
        stage[1] = torch.cuda.Event()
        stage[1].record()
        self.feed_ring.set_feed(self.plan_d[:n], self.B, regions)
        for r in recvs:
            r.wait()
        self.stats["local_batches"] += owners.count(me)
        self.stats["remote_batches"] += n - owners.count(me)
        by = self.stats.setdefault("remote_by_owner", {})
        for o in remote:
            by[o] = by.get(o, 0) + count[o]
        self.last_plan = plan

    def _capture_dp(self):
        """k updates of the data-parallel step as one graph.  Every cursor / RNG / optimizer state the launches read lives on
        the device and the sampler follows the step's plan from device memory, so a replay continues where the stream is;
        the graph begins with an explicit draw and ends with an update that draws nothing ahead (`last`).

        A capture that is refused part-way has run nothing on the device, but every launch call it recorded has advanced the
        learner's HOST-side launch state (which copy of the double-buffered optimizer state / dgrad images comes next, the
        armed noise request) and the stepper's input-set bookkeeping: both are snapshotted before the capture and put back
        on failure (`ddrl_sac1_capture_begin / _abort`), so the eager fallback continues bit for bit as if no capture had
        been tried.  DDRL_DP_CAPTURE_FAIL=i makes the i-th captured all-reduce raise (tests: i recorded apply() calls, one
        more recorded gradient pass)."""
        import os
        k = self.dp_per_graph
        fail_at = int(os.environ.get("DDRL_DP_CAPTURE_FAIL", "-1"))
        torch.cuda.synchronize()
        self.dp_grads.graph_sync()
        torch.cuda.synchronize()
        stream = torch.cuda.current_stream()
        self.learner.capture_begin()
        g = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(g):
                for i in range(k):
                    self.dp_grads(last=(i == k - 1))
                    if i == fail_at:
                        raise RuntimeError("injected capture failure before all-reduce %d (DDRL_DP_CAPTURE_FAIL)" % i)
                    comm.allreduce_mean_(self.dp_g, group=self.lgroup)
                    self.dp_apply()
                self.dp_grads.graph_sync()
        except Exception:
            torch.cuda.set_stream(stream)            # (a capture_end that raises leaves torch on its capture stream)
            torch.cuda.synchronize()
            self.learner.capture_abort()
            self.dp_grads.reset()
            raise
        self.dp_graph = g

    def _train_eager(self, n, end_of_step=False):
        """Several learners: per update, the plan's batch -> gradients -> all-reduce (mean) -> Adam + polyak; whole graphs of
        dp_per_graph updates where they fit, the rest eagerly."""
        k = self.dp_per_graph
        if k > 0 and n >= k:
            if self.dp_graph is None:
                self.dp_grads(last=True)                 # one eager update first: argument / empty-ring errors surface outside the capture
                comm.allreduce_mean_(self.dp_g, group=self.lgroup)
                self.dp_apply()
                n -= 1
                # The capture (k updates with their RCCL all-reduce nodes) has only ever run on a world-size-1 group here.  Should a real
                # multi-GPU group refuse it, the learners agree on that and fall back to the eager step instead of losing the run.
                ok = 1.0
                try:
                    self._capture_dp()
                except Exception as e:  # noqa
                    import sys
                    print("partition: capturing the data-parallel step failed (%r): eager updates from here on" % (e,), file=sys.stderr)
                    ok = 0.0                             # (_capture_dp has put the learner's and the stepper's host state back)
                flag = torch.tensor([ok], dtype=torch.float32, device=self.device)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.lgroup)
                if float(flag.item()) < 0.5:
                    self.dp_graph, self.dp_per_graph, k = None, 0, 0
            if k > 0:
                self.dp_grads.graph_sync()
            while k > 0 and n >= k:
                self.dp_graph.replay()
                n -= k
        for i in range(n):   # (every call ends with nothing drawn ahead: the ring / the plan / a graph may come next)
            self.dp_grads(last=(i == n - 1))
            comm.allreduce_mean_(self.dp_g, group=self.lgroup)
            self.dp_apply()

    def _free_front(self, n, t):
        """Free mode, everything in front of a step's updates: adopt the previous step's last push, K vector steps, the blocks of
        the NEXT step drawn and sent, this step's (already posted) blocks armed, the next step's receives posted."""
        first = self.pending is None and not getattr(self, "_free_started", False)
        if first:
            # prologue: this step's blocks travel now (drawn in front of the env steps: the rings as constructed), like a lock-step step
            self._free_started, self._free_n = True, n
            plans0 = [self.schedule.next() for _ in range(n)]
            if self.rb is not None:
                self._serve(plans0)
            if self.learner is not None:
                self.pending = self._post(plans0, self.regions_free, 0)
            self._free_parity = 0
        assert n == self._free_n, "free mode draws the next step's batches ahead: the updates per step must not change"
        if self.roll is not None:
            self._adopt_pushed()
            self.roll.step(self.free_steps)
        t = self._tick("s_env", t)
        plans_next = [self.schedule.next() for _ in range(n)]
        if self.rb is not None:
            self._serve(plans_next)                     # behind this step's env steps: what step s + 1 trains on
        t = self._tick("s_serve", t)
        if self.learner is not None:
            cur = self.pending
            self._free_parity ^= 1
            self.pending = self._post(plans_next, self.regions_free, self._free_parity)   # posted BEFORE this step's updates run
            self._arm(cur)
        t = self._tick("s_receive", t)
        return t

    def _tick(self, key, t0):
        """DDRL_PART_TIMING=1: wall time per phase (device drained at every phase boundary — diagnosis only)."""
        if not self.timing:
            return 0.0
        import time
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        if key is not None:
            self.stats[key] = self.stats.get(key, 0.0) + (t1 - t0)
        return t1

    def _poll_error(self, drain=False):
        """The sampler's sticky device-side error (a plan entry out of range, a local draw from an empty ring) of the steps
        whose updates have finished — one step behind the host, so that the device keeps running ahead."""
        while self.err_stage and (drain or len(self.err_stage) >= PLAN_STAGES - 1 or self.err_stage[0][1].query()):
            word, ev = self.err_stage.pop(0)
            ev.synchronize()
            rc = int(word[0])
            if rc != 0:
                raise ValueError("replay sampler error %d on rank %d: %s" % (rc, self.roles.rank,
                                 "local draw from an empty ring (high <= 0)" if rc == -2 else "a feed-plan entry was out of range; that update trained on a stale input set"))

    def check(self):
        """Drain the device and raise what its samplers reported."""
        if self.pending is not None:                 # free mode: the blocks posted for the next step (their sends have been issued) land
            for r in self.pending["recvs"]:          # now; a later step() arms them as usual, a run that ends here leaves nothing in flight
                r.wait()
            self.pending["recvs"] = []
        if self.learner is not None:
            self._poll_error(drain=True)

    def step(self, n_updates):
        if len(self.step_events) >= PLAN_STAGES:
            self.step_events.pop(0).synchronize()     # the step issued PLAN_STAGES steps ago has left the device
        if self.learner is not None:
            self._poll_error()
        t = self._tick(None, 0.0)
        n = int(n_updates)
        if self.free_steps > 0:
            t = self._free_front(n, t)
        else:
            if self.roll is not None:
                self.roll.step()
            t = self._tick("s_env", t)
            plans = [self.schedule.next() for _ in range(n)]
            if self.rb is not None:
                self._serve(plans)
            t = self._tick("s_serve", t)
            if self.learner is not None:
                self._receive(plans)
            t = self._tick("s_receive", t)
        left = n
        while left > 0:
            seg = min(left, self.push_freq - self.cnt % self.push_freq)   # updates until the next push
            if self.learner is not None:
                if self.loop is not None:
                    self.loop.run(seg)
                else:
                    self._train_eager(seg, end_of_step=(left == seg))
            t = self._tick("s_updates", t)
            self.cnt += seg
            left -= seg
            if self.cnt % self.push_freq == 0 and self.bcast is not None:
                self._push()
                t = self._tick("s_push", t)
        for w in self.sends:
            w.wait()
        self.sends = []
        if self.learner is not None:
            self.feed_ring.take_error(self.err_d)
            word = torch.empty(1, dtype=torch.int32).pin_memory()
            word.copy_(self.err_d, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self.err_stage.append((word, ev))
        ev = torch.cuda.Event()
        ev.record()
        self.step_events.append(ev)
        self._tick("s_drain", t)
