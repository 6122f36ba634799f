"""The package's lazy export table (PEP 562 module __getattr__), shared by `distributed-drl_amd/__init__.py` and by the importable
alias package `distributed_drl_amd/` (the directory name the layout prescribes is not a Python identifier)."""
__version__ = "0.1.0"

_TABLE = {
    "ReplayBuffer": "replay", "ReplayBufferSAC1": "replay", "ReplayBufferDQN": "replay", "ReplayBufferNStep": "replay",
    "ParameterServer": "ps", "ParameterServerNode": "ps",
    "Learner": "agent", "Actor": "agent", "Model": "agent", "HyperParameters": "agent",
    "VecLunarLander": "env",
    "worker_rollout": "workers", "worker_train": "workers", "worker_test": "workers",
    "worker_rollout_sac1": "workers", "worker_train_sac1": "workers", "worker_test_sac1": "workers",
    "worker_rollout_dqn": "workers", "worker_train_dqn": "workers", "worker_test_dqn": "workers", "BatchCache": "workers", "get_al_status": "workers",
    "worker_rollout_nstep": "workers", "worker_train_nstep": "workers",
    "RolloutDevice": "workers", "TrainDevice": "workers", "TrainDeviceDQN": "workers", "RolloutDeviceNStep": "workers", "WindowQueue": "workers", "ActorLearnerLoop": "workers", "FreeRunningLoop": "workers",
}


def make_getattr(package):
    """-> the module __getattr__ of `package`: heavy submodules import torch, so they load on first use."""
    def __getattr__(name):
        import importlib
        if name in _TABLE:
            return getattr(importlib.import_module("." + _TABLE[name], package), name)
        raise AttributeError(name)
    return __getattr__
