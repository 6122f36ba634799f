"""One-off control for tests/_rccl_world1_child.py's capture-fallback case: the same injected aborts with the restore switched off.
Expected: at least the i = 1 case (an odd number of recorded optimizer steps) DIFFERS from the never-captured run — i.e. the test
does detect the bug ADVICE r5 described."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", DDRL_DIST_FORCE="1", MASTER_PORT="29533")
import numpy as np, torch, torch.distributed as dist
import distributed_drl_amd as d
from distributed_drl_amd import _lib, comm, partition
from distributed_drl_amd.agent import HyperParameters, Learner
from distributed_drl_amd.workers import RolloutDevice
comm.init_from_env(); _lib.require_gpu()
opt = HyperParameters()
opt.num_envs, opt.batch_size, opt.seed, opt.start_steps, opt.max_ep_len, opt.push_freq = 64, 32, 5, -1, 50, 6
def shard():
    rb = d.ReplayBufferSAC1(opt.obs_dim, opt.act_dim, 4096, seed=100)
    rs = np.random.RandomState(0); m = 500
    rb.store_batch(*(torch.from_numpy(x).cuda() for x in (rs.randn(m, 8).astype(np.float32), rs.uniform(-1, 1, (m, 2)).astype(np.float32),
                   np.arange(m, dtype=np.float32), rs.randn(m, 8).astype(np.float32), np.zeros(m, np.float32))))
    return rb
def run_it(dp_graph):
    run = partition.PartitionedRun(opt, partition.Roles(1, 0), shard, lambda rb: RolloutDevice(None, rb, opt, worker_index=0),
                                   lambda: Learner(opt, job="learner", index=0), seed=9, updates_per_graph=0, force_dp=True, dp_updates_per_graph=dp_graph)
    for n in (3, 4, 4, 4, 5):
        run.step(n)
    torch.cuda.synchronize(); run.check()
    return run.learner.get_weights_flat().cpu().numpy()
ref = run_it(0)
Learner.capture_abort = lambda self: None     # the restore switched off: round 5's fallback
for i in (0, 1, 2):
    os.environ["DDRL_DP_CAPTURE_FAIL"] = str(i)
    w = run_it(4)
    print("restore OFF, abort before all-reduce %d: %s (max |diff| %.3g)" % (i, "identical" if np.array_equal(w, ref) else "DIFFERS", np.abs(w - ref).max()), flush=True)
dist.destroy_process_group()
