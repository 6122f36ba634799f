"""Multi-GPU plumbing over torch.distributed (backend "nccl" == RCCL over xGMI on ROCm; "gloo" for
the CPU tests).  One process per GPU, launched by torch.distributed.run.

Mapping of the reference's cross-process traffic:
  * ps.push / ps.pull (example/dsac.py:59-65, every 300 updates / every episode)
        -> ONE broadcast of the flat float32 parameter vector from the learner rank
           (`ParamBroadcast`).  1.5 MB: latency-bound, so it is a single collective, never 20.
  * replay shards (algos/sac1/sac_ray.py:137-141,246; algos/dqn/train.py:191-199,277-279):
        every rank appends to its LOCAL shard (no collective on store); a sampled batch comes
        from ONE shard chosen as np.random.choice(num_shards) on the learner's seeded stream
        (`ShardPicker`); the owner draws all the batches it owes a learner for one step in one
        launch sequence and sends them as ONE point-to-point block (partition.py: `_serve` /
        `_receive`, RCCL send/recv) — no per-batch message, no broadcast.
  * 2 learners (BASELINE config 4) -> all-reduce(sum)/k of the flat gradient between learner ranks
        (`allreduce_mean_`); the reference's multi-learner is unsynchronised last-writer-wins
        (example/dsac.py:59-62,233), so this is a documented new synchronous semantics.
The path shards by independent units (envs, replay shards): bench.py reports weak scaling.
"""
import datetime
import os

import numpy as np
import torch
import torch.distributed as dist


def init_from_env(backend=None, force=None):
    """Join the process group described by RANK / WORLD_SIZE / MASTER_* (no-op for 1 process, unless `force` /
    DDRL_DIST_FORCE=1 asks for a world-size-1 group: the same RCCL code path as N > 1 on a one-GPU box)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if force is None:
        force = os.environ.get("DDRL_DIST_FORCE", "0") == "1"
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            # DDRL_DIST_BACKEND=gloo lets several ranks share ONE GPU for functional checks
            # (RCCL refuses duplicate devices); production = "nccl" (RCCL over xGMI)
            backend = os.environ.get("DDRL_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        # a collective that never completes (a peer died, a mismatched send / recv) aborts after DDRL_DIST_TIMEOUT_S instead of
        # the backend's 10-minute default; binding the group to its device makes RCCL build the communicator now, so the first
        # isend / irecv of a run does not create two-rank communicators lazily in an order the peers may not share
        kw = {"timeout": datetime.timedelta(seconds=int(os.environ.get("DDRL_DIST_TIMEOUT_S", "300")))}
        if backend == "nccl":
            dev = local % max(1, torch.cuda.device_count())
            torch.cuda.set_device(dev)
            kw["device_id"] = torch.device("cuda", dev)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, world, local


def world_size():
    return dist.get_world_size() if dist.is_initialized() else 1


def rank():
    return dist.get_rank() if dist.is_initialized() else 0


def barrier():
    if dist.is_initialized():
        dist.barrier()


def allreduce_mean_(flat, group=None):
    """In-place mean over the group's ranks (learner gradient all-reduce)."""
    if not dist.is_initialized():
        return flat
    if dist.get_backend(group) == "nccl":      # RCCL averages inside the reduction kernel
        dist.all_reduce(flat, op=dist.ReduceOp.AVG, group=group)
        return flat
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    n = dist.get_world_size(group)
    if n > 1:
        flat.div_(n)
    return flat


def allreduce_max(value, device=None):
    """MAX over ranks of a python float (bench timing)."""
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64,
                     device=device if device is not None else ("cuda" if dist.get_backend() == "nccl" else "cpu"))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def allreduce_sum(value, device=None):
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64,
                     device=device if device is not None else ("cuda" if dist.get_backend() == "nccl" else "cpu"))
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


class ParamBroadcast:
    """ps.push + ps.pull across ranks: the learner rank's flat parameter vector is broadcast into
    every rank's receive buffer; `version` counts pushes (ranks call `sync` collectively at the
    same cadence — every push_freq updates, algos/sac1/sac1.py:149)."""

    def __init__(self, count, device, src=0, group=None):
        self.buf = torch.zeros(int(count), dtype=torch.float32, device=device)
        self.src, self.group, self.version = src, group, 0

    def sync(self, flat=None):
        """Collective.  On the source rank `flat` is the fresh parameter vector."""
        if rank() == self.src:
            assert flat is not None and flat.numel() == self.buf.numel()
            self.buf.copy_(flat.reshape(-1))
        if dist.is_initialized():
            dist.broadcast(self.buf, src=self.src, group=self.group)
        self.version += 1
        return self.buf

    # the slice of ParameterServer's surface the device workers use
    def pull_flat(self, offset, count, out=None):
        v = self.buf[int(offset):int(offset) + int(count)]
        if out is not None:
            out.copy_(v)
            return out
        return v.clone()

    def span(self, keys):
        return getattr(self, "_spans", {}).get(tuple(keys))

    def register_span(self, keys, offset, count):
        self.__dict__.setdefault("_spans", {})[tuple(keys)] = (int(offset), int(count))


class ShardPicker:
    """Shard choice of the sharded sampler: `np.random.choice(num_shards, 1)[0]` in the reference
    (algos/sac1/sac_ray.py:137; algos/dqn/train.py:191-193), which consumes the legacy MT19937
    stream exactly like randint(0, num_shards) (SURVEY §7.2 probe).  Every rank holds the same
    seeded stream, so all ranks agree on the owner without communication."""

    def __init__(self, num_shards, seed=0):
        self.num_shards = int(num_shards)
        self.rs = np.random.RandomState(int(seed) & 0xFFFFFFFF)

    def next(self):
        return int(self.rs.randint(0, self.num_shards))
