"""distributed-drl_amd — MI355X-native actor–learner hot path with the Distributed-DRL surface.

Package contents (only what the path needs):
  csrc/         HIP kernels + the C-ABI (include/ddrl.h) -> libddrl_hip.so
  _lib          ctypes binding (fails loudly when the library or the GPU is missing)
  replay        ReplayBuffer / ReplayBufferSAC1 / ReplayBufferDQN   (example/dsac.py:14-48 ...)
  ps            ParameterServer                                      (example/dsac.py:51-73)
  agent         Learner / Actor (SAC1)                               (algos/sac1/actor_learner.py)
  env           VecLunarLander (batched env.step/reset)              (gym call sites dsac.py:78-127)
  workers       worker_rollout / worker_train / worker_test          (example/dsac.py:76-177)
  remote        the slice of Ray's API the drivers use (remote/get/wait)
  comm          torch.distributed (RCCL) plumbing for ps.push/pull and replay shards
"""
from ._exports import __version__, make_getattr  # noqa: F401
from . import _lib  # noqa: F401
from . import remote  # noqa: F401

__getattr__ = make_getattr(__name__)
