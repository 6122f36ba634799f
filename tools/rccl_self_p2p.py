"""Probe: does RCCL accept a grouped send/recv to self at world size 1?  If yes, partition._send / partition._Recv (the block
transfer of configs 3/4) run once under backend nccl on the one-GPU box.  Run under `timeout`: a lone self-send would block."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", DDRL_DIST_FORCE="1")
os.environ.setdefault("MASTER_PORT", "29519")
os.environ.pop("DDRL_DIST_BACKEND", None)
import torch
import torch.distributed as dist
from torch.distributed.distributed_c10d import _coalescing_manager
from distributed_drl_amd import comm, partition

comm.init_from_env()
dev = torch.device("cuda", torch.cuda.current_device())
src = torch.arange(5120 * 64, dtype=torch.float32, device=dev)
dst = torch.zeros_like(src)
mode = sys.argv[1] if len(sys.argv) > 1 else "batch"
if mode == "batch":
    works = dist.batch_isend_irecv([dist.P2POp(dist.isend, src, 0), dist.P2POp(dist.irecv, dst, 0)])
    for w in works:
        w.wait()
else:
    with _coalescing_manager(device=dev, async_ops=True) as cm:
        partition._send(src, 0)
        r = partition._Recv(dst, 0)
    cm.wait()
torch.cuda.synchronize()
print("SELF_P2P", mode, "ok" if torch.equal(src, dst) else "MISMATCH", flush=True)
dist.destroy_process_group()
