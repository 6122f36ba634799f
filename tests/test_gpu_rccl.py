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
    # an aborted capture of the data-parallel step (after 0 / an odd / an even number of recorded optimizer steps) falls back to
    # eager updates that equal, bit for bit, a run that never captured (ddrl_sac1_capture_begin / _abort)
    assert "rccl capture fallback ok" in out


@pytest.mark.parametrize("mode", ["plain", "torch"])
def test_comm_cabi_world1(mode):
    """include/ddrl.h's comm_* (last row) from a process that binds libddrl_hip.so with ctypes alone, and from one that
    already holds PyTorch's RCCL: unique id, init, parameter broadcast, gradient all-reduce (mean), a block sent to self inside a
    group, argument checking — and exactly one RCCL mapped in the process either way."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run([sys.executable, os.path.join(HERE, "_comm_cabi_child.py"), mode], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       timeout=300)
    out = p.stdout.decode("utf-8", "replace")
    assert p.returncode == 0 and "COMM_CABI_OK " + mode in out, out[-4000:]
