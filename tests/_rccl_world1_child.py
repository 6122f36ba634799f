"""Child of tests/test_gpu_rccl.py (a fresh process: the process group must be created before anything else touches the
GPU).  World size 1 on backend "nccl" (= RCCL): every torch.distributed call the N > 1 runs make — init, broadcast,
all-reduce AVG, barrier, new_group, and PartitionedRun's own step — goes through RCCL here, with no gloo branch taken.
The reference traffic these replace: ps.push / ps.pull (example/dsac.py:59-65), the per-shard sample RPC
(algos/sac1/sac_ray.py:137-141)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", DDRL_DIST_FORCE="1")
os.environ.setdefault("MASTER_PORT", "29517")
os.environ.pop("DDRL_DIST_BACKEND", None)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import distributed_drl_amd as d  # noqa: E402
from distributed_drl_amd import _lib, comm, partition  # noqa: E402
from distributed_drl_amd.agent import HyperParameters, Learner  # noqa: E402
from distributed_drl_amd.workers import RolloutDevice  # noqa: E402


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    r, w, local = comm.init_from_env()
    assert (r, w) == (0, 1) and dist.is_initialized() and dist.get_backend() == "nccl", dist.get_backend()
    assert not partition._is_gloo()
    _lib.require_gpu()
    dev = torch.device("cuda", torch.cuda.current_device())

    # ---- comm.py over RCCL ---------------------------------------------------------------------
    n = 375106
    pb = comm.ParamBroadcast(n, dev, src=0)
    flat = torch.arange(n, dtype=torch.float32, device=dev) * 0.25
    assert torch.equal(pb.sync(flat), flat) and pb.version == 1
    g = torch.randn(n, device=dev)
    g0 = g.clone()
    comm.allreduce_mean_(g)                                    # ReduceOp.AVG inside RCCL
    assert torch.equal(g, g0)
    grp = dist.new_group(ranks=[0])
    comm.allreduce_mean_(g, group=grp)
    assert torch.equal(g, g0)
    comm.barrier()
    assert comm.allreduce_max(3.5, device=dev) == 3.5 and comm.allreduce_sum(2.0, device=dev) == 2.0
    # ---- the block transfer of configs 3/4 (partition._send / partition._Recv: RCCL send / recv) -----------------------------
    # world size 1 has no remote owner; RCCL accepts a send to self when the matching receive sits in the same group call, so
    # one block of 64 packed batches goes owner -> learner through the very functions PartitionedRun uses
    from torch.distributed.distributed_c10d import _coalescing_manager
    nf = partition.batch_floats(8, 2, 256)
    blk = torch.randn(64 * nf, device=dev)
    region = torch.zeros_like(blk)
    with _coalescing_manager(device=dev, async_ops=True) as cm:
        partition._send(blk, 0)
        rcv = partition._Recv(region, 0)
    cm.wait()
    assert rcv.stage is None                                   # device buffers straight into RCCL: no host staging
    torch.cuda.synchronize()
    assert torch.equal(region, blk)
    print("rccl comm ok", flush=True)
    if what == "comm":
        return

    # ---- PartitionedRun at world 1 through the N > 1 code path --------------------------------
    opt = HyperParameters()
    opt.num_envs, opt.batch_size, opt.seed, opt.start_steps, opt.max_ep_len, opt.push_freq = 64, 32, 5, -1, 50, 6

    def shard():
        rb = d.ReplayBufferSAC1(opt.obs_dim, opt.act_dim, 4096, seed=100)
        rs = np.random.RandomState(0)
        m = 500
        rb.store_batch(*(torch.from_numpy(x).cuda() for x in (
            rs.randn(m, 8).astype(np.float32), rs.uniform(-1, 1, (m, 2)).astype(np.float32), np.arange(m, dtype=np.float32),
            rs.randn(m, 8).astype(np.float32), np.zeros(m, np.float32))))
        return rb

    def run_it(force_dp, per_graph, dp_graph=0, capture_fails=False):
        roles = partition.Roles(1, 0)
        run = partition.PartitionedRun(opt, roles, shard, lambda rb: RolloutDevice(None, rb, opt, worker_index=0),
                                       lambda: Learner(opt, job="learner", index=0), seed=9, updates_per_graph=per_graph, force_dp=force_dp,
                                       dp_updates_per_graph=dp_graph)
        assert run.bcast is not None and (run.lgroup is not None) == force_dp
        if force_dp:
            assert run.dp_per_graph == dp_graph
        for n in (3, 4, 4, 4, 5):                              # 20 updates: pushes (RCCL broadcasts) at 6, 12, 18 + the initial one.  The
            run.step(n)                                        # first capture attempt comes after 3 + 1 eager updates: optimizer state on copy 0, copy 1 one update OLDER
        torch.cuda.synchronize()
        run.check()
        if dp_graph and not capture_fails:
            assert run.dp_graph is not None                    # steps 3 and 4 are whole graphs of 4 updates, step 5 a graph + 1 eager
        if capture_fails:
            assert run.dp_graph is None and run.dp_per_graph == 0   # the learners agreed on the eager step
        assert run.learner.opt_steps() == (20, 20) and run.stats["pushes"] == 4
        assert run.rb.get_counts() == (20, 500 + 5 * 64, 500 + 5 * 64)
        n_pi = run.roll.actor.n_params
        assert torch.equal(run.roll.actor.get_weights_flat(), run.bcast.buf[:n_pi])     # the rollout runs push #3's policy
        return run.learner.get_weights_flat().cpu().numpy()

    w_loop = run_it(False, 2)          # single learner: graph-captured loop + broadcast
    w_dp = run_it(True, 0)             # data-parallel step with a group of one: gradients -> RCCL all-reduce AVG -> apply
    np.testing.assert_array_equal(w_loop, w_dp)
    w_dpg = run_it(True, 0, dp_graph=4)   # the same step captured with its RCCL all-reduce as graphs of 4 updates
    np.testing.assert_array_equal(w_loop, w_dpg)
    # ---- the capture fallback (partition.py: _capture_dp / _train_eager) ------------------------------------------------------
    # A capture refused after i recorded apply() calls (and i + 1 recorded gradient passes) has advanced the learner's host-side
    # launch state — which optimizer-state copy / dgrad image the next launch reads, the armed noise request — without the
    # device running anything.  The eager fallback must continue bit for bit like a run that never tried to capture: i = 0
    # (only a gradient pass recorded), i = 1 (an ODD number of recorded optimizer steps: the stale-copy case — the capture is tried
    # after an even number of eager updates, so the two copies of the optimizer state DIFFER at that point; with the restore
    # switched off this case ends on different weights, tools/capture_fallback_control.py), i = 2 (even).
    for i in (0, 1, 2):
        os.environ["DDRL_DP_CAPTURE_FAIL"] = str(i)
        try:
            w_fb = run_it(True, 0, dp_graph=4, capture_fails=True)
        finally:
            os.environ.pop("DDRL_DP_CAPTURE_FAIL", None)
        np.testing.assert_array_equal(w_loop, w_fb)
    print("rccl capture fallback ok", flush=True)
    print("rccl partition ok", flush=True)


if __name__ == "__main__":
    main()
    if dist.is_initialized():
        dist.destroy_process_group()
    print("RCCL_WORLD1_OK", flush=True)
