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
__version__ = "0.1.0"

from . import _lib  # noqa: F401
from . import remote  # noqa: F401


def __getattr__(name):
    # heavy submodules import torch; load them on first use
    import importlib
    table = {
        "ReplayBuffer": "replay", "ReplayBufferSAC1": "replay", "ReplayBufferDQN": "replay", "ReplayBufferNStep": "replay",
        "ParameterServer": "ps", "ParameterServerNode": "ps",
        "Learner": "agent", "Actor": "agent", "Model": "agent", "HyperParameters": "agent",
        "VecLunarLander": "env",
        "worker_rollout": "workers", "worker_train": "workers", "worker_test": "workers",
        "worker_rollout_sac1": "workers", "worker_train_sac1": "workers", "worker_test_sac1": "workers",
        "worker_rollout_dqn": "workers", "worker_train_dqn": "workers", "worker_test_dqn": "workers", "BatchCache": "workers", "get_al_status": "workers",
        "worker_rollout_nstep": "workers", "worker_train_nstep": "workers",
        "RolloutDevice": "workers", "TrainDevice": "workers", "TrainDeviceDQN": "workers", "RolloutDeviceNStep": "workers", "WindowQueue": "workers", "ActorLearnerLoop": "workers",
    }
    if name in table:
        return getattr(importlib.import_module("." + table[name], __name__), name)
    raise AttributeError(name)
