"""Child of tests/test_gpu_rccl.py::test_comm_cabi_*: the comm_* entry points of libddrl_hip.so at world size 1, bound with ctypes
ALONE (no torch import: device memory through hipMalloc of libamdhip64) — what a non-Python host of the library would do — or, with
argument "torch", in a process that already holds PyTorch's RCCL (the library must share that copy, not load a second one)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
mode = sys.argv[1] if len(sys.argv) > 1 else "plain"
if mode == "torch":
    import torch
    torch.cuda.init()
hip = ctypes.CDLL("libamdhip64.so")
lib = ctypes.CDLL(os.path.join(ROOT, "distributed-drl_amd", "libddrl_hip.so"))
lib.ddrl_last_error.restype = ctypes.c_char_p
P = ctypes.c_void_p


def ok(rc):
    assert rc == 0, (rc, lib.ddrl_last_error())


def dmalloc(nbytes):
    p = P()
    assert hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(nbytes)) == 0
    return p


n = 375106
host = (ctypes.c_float * n)(*[0.25 * (i % 1000) for i in range(n)])
back = (ctypes.c_float * n)()
a, b = dmalloc(4 * n), dmalloc(4 * n)
assert hip.hipMemcpy(a, host, ctypes.c_size_t(4 * n), 1) == 0
uid = (ctypes.c_uint8 * 128)()
ok(lib.ddrl_comm_unique_id(uid))
assert any(uid)
h = P()
ok(lib.ddrl_comm_init(ctypes.byref(h), 0, 0, 1, uid))
ok(lib.ddrl_comm_bcast_params(h, a, ctypes.c_int64(n), 0, None))          # ps.push + pull
ok(lib.ddrl_comm_allreduce_grads(h, a, ctypes.c_int64(n), None))          # mean over one rank: identity
ok(lib.ddrl_comm_group_start())                                           # a block to self: send + matching receive in one group
ok(lib.ddrl_comm_send_batch(h, a, ctypes.c_int64(n), 0, None))
ok(lib.ddrl_comm_recv_batch(h, b, ctypes.c_int64(n), 0, None))
ok(lib.ddrl_comm_group_end())
assert hip.hipDeviceSynchronize() == 0
assert hip.hipMemcpy(back, b, ctypes.c_size_t(4 * n), 2) == 0
assert bytes(back) == bytes(host)
assert lib.ddrl_comm_bcast_params(h, a, ctypes.c_int64(n), 3, None) == -1     # root outside the communicator: DDRL_ERR_BAD_ARG
ok(lib.ddrl_comm_destroy(h))
maps = open("/proc/self/maps").read()
copies = sorted({line.split()[-1] for line in maps.splitlines() if "librccl" in line})
assert len(copies) == 1, copies                                            # exactly one RCCL in the process
if mode == "torch":
    assert "torch" in copies[0], copies
print("COMM_CABI_OK", mode, copies[0], flush=True)
