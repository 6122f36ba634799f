"""ctypes binding of libddrl_hip.so (the C-ABI declared in include/ddrl.h).

The product path has no CPU fallback: if the shared library is missing, or a call fails, this
module raises.  Nothing here imports the oracle."""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_uint8, c_uint32, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# DDRL_LIB_PATH: another build of the same C-ABI (same-box A/B runs of tools/*: an older library next to the in-tree one, selected
# per process instead of copied over the package file).  Symbols an older build lacks are skipped (their call sites then fail).
LIB_PATH = os.environ.get("DDRL_LIB_PATH") or os.path.join(_HERE, "libddrl_hip.so")

DDRL_OK = 0
DDRL_ERR_BAD_ARG = -1
DDRL_ERR_EMPTY_BUFFER = -2
DDRL_ERR_HIP = -3
DDRL_ERR_NOMEM = -4
DDRL_ERR_UNSUPPORTED = -5
DDRL_ERR_NOT_REPRESENTABLE = -6
DDRL_ERR_RCCL = -7
DDRL_REPLAY_ACTS_1D = 1
DDRL_REPLAY_U8_OBS = 2
DDRL_ENV_STATE_FIELDS = 32
SAC1_STAGES = 12
DQN_STAGES = 9   # include/ddrl.h: DDRL_DQN_STAGES
SAC1_MAIN, SAC1_TARGET, SAC1_ADAM_M, SAC1_ADAM_V, SAC1_GRAD = range(5)
SAC1, SAC_V = 0, 1  # ddrl_sac1_config_t.variant


class Sac1Config(ctypes.Structure):
    """ddrl_sac1_config_t; defaults = algos/sac1/hyperparams.py + core.py:91 at LunarLander dims."""
    _fields_ = [("obs_dim", c_int32), ("act_dim", c_int32), ("hidden1", c_int32), ("hidden2", c_int32),
                ("batch", c_int32), ("variant", c_int32), ("alpha", c_double), ("gamma", c_double),
                ("lr", c_double), ("polyak", c_double), ("beta1", c_double), ("beta2", c_double),
                ("adam_eps", c_double), ("act_scale", c_double)]

    def __init__(self, obs_dim=8, act_dim=2, hidden1=400, hidden2=300, batch=256, alpha=0.1, gamma=0.997,
                 lr=5e-5, polyak=0.995, beta1=0.9, beta2=0.999, adam_eps=1e-8, act_scale=1.0, variant=0):
        super().__init__(obs_dim, act_dim, hidden1, hidden2, batch, variant, alpha, gamma, lr, polyak, beta1, beta2,
                         adam_eps, act_scale)


class DqnConfig(ctypes.Structure):
    """ddrl_dqn_config_t; defaults = algos/dqn/hyperparams.py."""
    _fields_ = [("obs_dim", c_int32), ("n_actions", c_int32), ("hidden1", c_int32), ("hidden2", c_int32), ("batch", c_int32),
                ("variant", c_int32), ("gamma", c_double), ("lr", c_double), ("polyak", c_double), ("beta1", c_double),
                ("beta2", c_double), ("adam_eps", c_double), ("alpha", c_double)]

    def __init__(self, obs_dim, n_actions, hidden1=400, hidden2=300, batch=128, gamma=0.99, lr=1e-3, polyak=0.995, beta1=0.9,
                 beta2=0.999, adam_eps=1e-8, variant=0, alpha=0.1):
        super().__init__(obs_dim, n_actions, hidden1, hidden2, batch, variant, gamma, lr, polyak, beta1, beta2, adam_eps, alpha)


_P = c_void_p  # device pointers and opaque handles cross as void*

# name -> (restype, argtypes).  Must list every symbol include/ddrl.h declares
# (tests/test_boundary.py checks header <-> table <-> library).
SIGNATURES = {
    "ddrl_version": (c_int, []),
    "ddrl_last_error": (c_char_p, []),
    "ddrl_device_arch": (c_int, [c_int, c_char_p, c_int]),
    "ddrl_host_device_pointer": (c_int, [_P, ctypes.POINTER(ctypes.c_void_p)]),
    "ddrl_replay_create": (c_int, [POINTER(_P), c_int, c_int64, c_int, c_int, c_uint32]),
    "ddrl_replay_destroy": (c_int, [_P]),
    "ddrl_replay_seed": (c_int, [_P, c_uint32, _P]),
    "ddrl_replay_store": (c_int, [_P, _P, _P, _P, _P, _P, c_int64, _P]),
    "ddrl_replay_sample": (c_int, [_P, c_int64, _P, _P, _P, _P, _P, _P, _P]),
    "ddrl_replay_gather": (c_int, [_P, _P, c_int64, _P, _P, _P, _P, _P, _P]),
    "ddrl_replay_create_ex": (c_int, [POINTER(_P), c_int, c_int64, c_int32, POINTER(c_int32), c_int64, c_int64]),
    "ddrl_replay_store_ex": (c_int, [_P, POINTER(_P), c_int64, _P]),
    "ddrl_replay_store_masked_ex": (c_int, [_P, POINTER(_P), _P, c_int64, _P]),
    "ddrl_replay_sample_ex": (c_int, [_P, c_int64, POINTER(_P), _P, _P]),
    "ddrl_replay_sample_many": (c_int, [_P, c_int64, c_int64, POINTER(_P), _P]),
    "ddrl_replay_set_feed": (c_int, [_P, _P, c_int32, c_int32, c_int32, POINTER(_P), POINTER(c_int32), _P]),
    "ddrl_replay_take_error": (c_int, [_P, _P, _P]),
    "ddrl_replay_gather_ex": (c_int, [_P, _P, c_int64, POINTER(_P), _P]),
    "ddrl_replay_buffers_ex": (c_int, [_P, POINTER(_P), POINTER(c_int32), POINTER(c_int32)]),
    "ddrl_replay_counts": (c_int, [_P, POINTER(c_int64), POINTER(c_int64), POINTER(c_int64), POINTER(c_int64), _P]),
    "ddrl_replay_buffers": (c_int, [_P, POINTER(_P), POINTER(_P), POINTER(_P), POINTER(_P), POINTER(_P)]),
    "ddrl_replay_set_counts": (c_int, [_P, c_int64, c_int64, c_int64, c_int64, _P]),
    "ddrl_replay_sample_indices": (c_int, [_P, c_int64, _P, _P]),
    "ddrl_replay_create_typed": (c_int, [POINTER(_P), c_int, c_int64, c_int32, POINTER(c_int32), _P, c_int64, c_int64]),
    "ddrl_replay_rows_export": (c_int, [_P, c_int32, c_int64, c_int64, _P, _P]),
    "ddrl_replay_rows_import": (c_int, [_P, c_int32, c_int64, c_int64, _P, _P]),
    "ddrl_replay_mt_state": (c_int, [_P, _P, POINTER(c_int32), _P]),
    "ddrl_ps_create": (c_int, [POINTER(_P), c_int, c_int64]),
    "ddrl_ps_destroy": (c_int, [_P]),
    "ddrl_ps_push": (c_int, [_P, _P, c_int64, c_int64, _P]),
    "ddrl_ps_pull": (c_int, [_P, _P, c_int64, c_int64, _P]),
    "ddrl_ps_buffer": (c_int, [_P, POINTER(_P), POINTER(c_int64)]),
    "ddrl_ps_version": (c_int64, [_P]),
    "ddrl_sac1_param_counts": (c_int, [POINTER(Sac1Config), POINTER(c_int64), POINTER(c_int64)]),
    "ddrl_sac1_create": (c_int, [POINTER(_P), c_int, POINTER(Sac1Config)]),
    "ddrl_sac1_destroy": (c_int, [_P]),
    "ddrl_sac1_set_weights": (c_int, [_P, _P, _P]),
    "ddrl_sac1_get_weights": (c_int, [_P, _P, _P]),
    "ddrl_sac1_export": (c_int, [_P, c_int, _P, _P]),
    "ddrl_sac1_import": (c_int, [_P, c_int, _P, _P]),
    "ddrl_sac1_grad_buffer": (c_int, [_P, POINTER(_P), POINTER(c_int64)]),
    "ddrl_sac1_grad_finalize": (c_int, [_P, _P]),
    "ddrl_sac1_opt_steps": (c_int, [_P, POINTER(c_int64), POINTER(c_int64), _P]),
    "ddrl_sac1_step": (c_int, [_P] + [_P] * 12 + [_P]),
    "ddrl_sac1_compute_grads": (c_int, [_P] + [_P] * 12 + [_P]),
    "ddrl_sac1_apply_grads": (c_int, [_P, _P]),
    "ddrl_sac1_opt_state_get": (c_int, [_P, POINTER(c_int64), POINTER(c_int64), POINTER(c_uint64), _P]),
    "ddrl_sac1_opt_state_set": (c_int, [_P, c_int64, c_int64, c_uint64, _P]),
    "ddrl_sac1_step_and_sample": (c_int, [_P, c_int, _P, c_int, _P]),
    "ddrl_sac1_compute_grads_and_sample": (c_int, [_P, c_int, _P, c_int, _P]),
    "ddrl_sac1_apply_grads_and_sample": (c_int, [_P, _P, c_int, _P]),
    "ddrl_sac1_step_host": (c_int, [_P, _P, c_int64, c_uint32, c_uint64, _P, _P]),
    "ddrl_sac1_graph_sync": (c_int, [_P, _P]),
    "ddrl_sac1_capture_begin": (c_int, [_P]),
    "ddrl_sac1_capture_abort": (c_int, [_P]),
    "ddrl_sac1_input_buffers": (c_int, [_P, c_int, POINTER(_P)]),
    "ddrl_sac1_is_fused": (c_int, [_P]),
    "ddrl_sac1_batch": (c_int, [_P]),
    "ddrl_sac1_fill_noise": (c_int, [_P, c_uint32, _P]),
    "ddrl_sac1_stage_time": (c_int, [_P, c_int, c_int, POINTER(c_float), _P]),
    "ddrl_loop_create": (c_int, [POINTER(_P), _P, _P, c_int32, c_uint32]),
    "ddrl_loop_destroy": (c_int, [_P]),
    "ddrl_loop_run": (c_int, [_P, c_int64, _P]),
    "ddrl_actor_create": (c_int, [POINTER(_P), c_int, POINTER(Sac1Config), c_int64]),
    "ddrl_actor_destroy": (c_int, [_P]),
    "ddrl_actor_set_weights": (c_int, [_P, _P, _P]),
    "ddrl_actor_get_weights": (c_int, [_P, _P, _P]),
    "ddrl_actor_act": (c_int, [_P, _P, _P, c_int64, c_int, _P, _P]),
    "ddrl_actor_act_one": (c_int, [_P, _P, c_uint32, c_uint64, c_int, _P, _P]),
    "ddrl_env_create": (c_int, [POINTER(_P), c_int, c_int64, c_uint32, c_int32]),
    "ddrl_env_destroy": (c_int, [_P]),
    "ddrl_env_reset": (c_int, [_P, _P, _P, _P]),
    "ddrl_env_step": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P]),
    "ddrl_rollout_begin": (c_int, [_P, _P, _P]),
    "ddrl_rollout_step": (c_int, [_P, _P, _P, c_int32, c_uint32, c_uint64, c_int, _P, _P, _P]),
    "ddrl_env_step_wrapped": (c_int, [_P, _P, c_float, c_float, c_float, c_int32, c_int32, _P, _P, _P, _P, _P, _P]),
    "ddrl_env_stats": (c_int, [_P, POINTER(c_int64), POINTER(c_double), POINTER(c_int64), _P]),
    "ddrl_env_get_state": (c_int, [_P, _P, _P]),
    "ddrl_env_set_state": (c_int, [_P, _P, _P]),
    "ddrl_actor_versions_enable": (c_int, [_P, c_int32, _P]),
    "ddrl_actor_versions_state": (c_int, [_P, _P, _P, _P]),
    "ddrl_actor_versions_adopt": (c_int, [_P, _P, c_int64, _P]),
    "ddrl_actor_act_versioned": (c_int, [_P, _P, _P, c_int64, c_int, c_int32, _P, _P]),
    "ddrl_comm_unique_id": (c_int, [_P]),
    "ddrl_comm_init": (c_int, [POINTER(_P), c_int, c_int32, c_int32, _P]),
    "ddrl_comm_destroy": (c_int, [_P]),
    "ddrl_comm_bcast_params": (c_int, [_P, _P, c_int64, c_int32, _P]),
    "ddrl_comm_allreduce_grads": (c_int, [_P, _P, c_int64, _P]),
    "ddrl_comm_send_batch": (c_int, [_P, _P, c_int64, c_int32, _P]),
    "ddrl_comm_recv_batch": (c_int, [_P, _P, c_int64, c_int32, _P]),
    "ddrl_comm_group_start": (c_int, []),
    "ddrl_comm_group_end": (c_int, []),
    "ddrl_dqn_param_count": (c_int, [_P, POINTER(c_int64)]),
    "ddrl_dqn_create": (c_int, [POINTER(_P), c_int, _P]),
    "ddrl_dqn_destroy": (c_int, [_P]),
    "ddrl_dqn_set_weights": (c_int, [_P, _P, _P]),
    "ddrl_dqn_export": (c_int, [_P, c_int, _P, _P]),
    "ddrl_dqn_import": (c_int, [_P, c_int, _P, _P]),
    "ddrl_dqn_step": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "ddrl_dqn_step_timed": (c_int, [_P, _P, _P, _P, _P, _P, c_int, _P, _P]),
    "ddrl_dqn_step_ring": (c_int, [_P, _P, _P, _P, _P, _P]),
    "ddrl_dqn_q": (c_int, [_P, _P, c_int64, _P, _P]),
    "ddrl_winq_create": (c_int, [POINTER(_P), c_int, c_int64, c_int32, c_int32, c_int32, c_int32]),
    "ddrl_winq_destroy": (c_int, [_P]),
    "ddrl_winq_begin": (c_int, [_P, _P, _P, _P]),
    "ddrl_winq_push": (c_int, [_P, _P, _P, _P, _P, _P, _P]),
    "ddrl_winq_buffers": (c_int, [_P, POINTER(_P), POINTER(_P)]),
    "ddrl_normal_fill": (c_int, [_P, c_int64, c_uint32, c_uint64, _P]),
    "ddrl_uniform_fill": (c_int, [_P, c_int64, c_float, c_float, c_uint32, c_uint64, _P]),
}

_lib = None


class DdrlError(RuntimeError):
    pass


def load():
    """Load libddrl_hip.so; raise loudly when it is absent (no fallback path exists)."""
    global _lib
    if _lib is not None:
        return _lib
    # torch first: libddrl_hip.so must bind to the HIP runtime torch has loaded (one runtime per
    # process), otherwise device pointers and streams would belong to a different runtime instance
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise DdrlError(
            "libddrl_hip.so not found at %s — build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C distributed-drl_amd/csrc`; there is no CPU fallback for the product path" % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)  # AttributeError if the library lacks a declared symbol
        except AttributeError:
            if not os.environ.get("DDRL_LIB_PATH"):
                raise
            continue                 # an OLDER build selected for an A/B run: calls of what it lacks fail at the call site
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    """Map a ddrl_status to the reference's error behaviour (SURVEY §8(b) Errors)."""
    if rc == DDRL_OK:
        return
    msg = load().ddrl_last_error().decode("utf-8", "replace")
    if rc == DDRL_ERR_EMPTY_BUFFER:
        raise ValueError(msg or "high <= 0")  # what np.random.randint(0, 0, n) raises in the reference
    if rc == DDRL_ERR_BAD_ARG or rc == DDRL_ERR_NOT_REPRESENTABLE:
        raise ValueError("ddrl: " + msg)
    if rc == DDRL_ERR_NOMEM:
        raise MemoryError("ddrl: " + msg)
    raise DdrlError("ddrl error %d: %s" % (rc, msg))


def require_gpu():
    """Fail loudly unless torch sees a gfx950 device and the library is loadable."""
    import torch
    lib = load()
    if not torch.cuda.is_available():
        raise DdrlError("no GPU visible: the MI355X hot path has no CPU fallback")
    buf = ctypes.create_string_buffer(128)
    check(lib.ddrl_device_arch(torch.cuda.current_device(), buf, 128))
    arch = buf.value.decode()
    if not arch.startswith("gfx950"):
        raise DdrlError("libddrl_hip.so is built for gfx950 only; device reports %r" % arch)
    return arch


def stream_ptr():
    import torch
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def dptr(t):
    """Device pointer of a contiguous float32/int64/uint8 CUDA tensor (or None -> NULL)."""
    if t is None:
        return c_void_p(None)
    assert t.is_cuda and t.is_contiguous(), "expected a contiguous device tensor"
    return c_void_p(t.data_ptr())
