"""RCCL (torch.distributed backend "nccl") exercised on the one-GPU box: a world-size-1 process group driven through the
same comm.py / partition.py calls the multi-GPU configurations make.  Runs in a fresh child process (the group must
exist before anything else initialises the GPU; a process that has touched the GPU is never re-exec'd)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


def test_rccl_world1_comm_and_partitioned_run():
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run([sys.executable, os.path.join(HERE, "_rccl_world1_child.py")], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       timeout=600)
    out = p.stdout.decode("utf-8", "replace")
    assert p.returncode == 0 and "RCCL_WORLD1_OK" in out, out[-4000:]
    assert "rccl comm ok" in out and "rccl partition ok" in out
