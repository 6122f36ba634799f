// comm_*: the reference's cross-process traffic on this path as RCCL calls behind the C-ABI, for a host that binds
// libddrl_hip.so without PyTorch (last row).  What each entry replaces:
//   ddrl_comm_bcast_params     ps.push(keys, values) by the learner + ps.pull(keys) by every worker (example/dsac.py:59-65,
//                              algos/sac1/sac1.py:149): ONE broadcast of the flat parameter vector
//   ddrl_comm_allreduce_grads  (new synchronous semantics for num_learners > 1, example/dsac.py:233) mean of the flat gradient
//   ddrl_comm_send_batch /     the reply of `replay_buffer[i].sample_batch.remote()` (algos/sac1/sac_ray.py:137-141): the block of
//   ddrl_comm_recv_batch       batches a shard owner drew for a learner's step, point to point over xGMI
// RCCL is bound at run time (dlopen): a process that already holds an RCCL (PyTorch's own copy) shares it, so there is never a
// second communicator library in the process; a plain C host gets the system's librccl.so.1.  DDRL_RCCL_PATH overrides.
// The Python package keeps using torch.distributed (backend "nccl" = the same RCCL) — this file is the torch-free binding.
#include "ddrl_common.h"
#include <dlfcn.h>
#include <mutex>
#include <stdlib.h>
#include <string.h>

namespace {

// the slice of rccl.h this file needs (RCCL 2.x ABI: ncclUniqueId = 128 opaque bytes, passed by value)
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;
constexpr int k_ncclFloat32 = 7, k_ncclAvg = 4;

struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Broadcast)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    const char *why = "";
};

Rccl &rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *env = getenv("DDRL_RCCL_PATH");
        if (env && *env) r.lib = dlopen(env, RTLD_NOW | RTLD_GLOBAL);
        // a copy the process already holds (PyTorch links its own librccl.so) before anything is loaded anew
        const char *held[] = {"librccl.so", "librccl.so.1"};
        for (int i = 0; !r.lib && i < 2; ++i) r.lib = dlopen(held[i], RTLD_NOW | RTLD_NOLOAD);
        const char *fresh[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (int i = 0; !r.lib && i < 3; ++i) r.lib = dlopen(fresh[i], RTLD_NOW | RTLD_GLOBAL);
        if (!r.lib) { r.why = "librccl.so.1 not found (set DDRL_RCCL_PATH)"; return; }
#define BIND(field, sym) r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.lib, sym)); if (!r.field) { r.why = "RCCL symbol missing: " sym; r.lib = nullptr; return; }
        BIND(GetUniqueId, "ncclGetUniqueId") BIND(CommInitRank, "ncclCommInitRank") BIND(CommDestroy, "ncclCommDestroy")
        BIND(Broadcast, "ncclBroadcast") BIND(AllReduce, "ncclAllReduce") BIND(Send, "ncclSend") BIND(Recv, "ncclRecv")
        BIND(GroupStart, "ncclGroupStart") BIND(GroupEnd, "ncclGroupEnd") BIND(GetErrorString, "ncclGetErrorString")
#undef BIND
    });
    return r;
}

int nccl_fail(const char *what, ncclResult_t rc) {
    Rccl &r = rccl();
    ddrl::set_error("%s: RCCL error %d (%s)", what, rc, r.GetErrorString ? r.GetErrorString(rc) : "?");
    return DDRL_ERR_RCCL;
}
#define DDRL_RCCL_READY()                                                                   \
    Rccl &R = rccl();                                                                       \
    if (!R.lib) { ddrl::set_error("RCCL unavailable: %s", R.why); return DDRL_ERR_RCCL; }
#define DDRL_NCCL_CHECK(call, what) do { const ncclResult_t rc_ = (call); if (rc_ != 0) return nccl_fail(what, rc_); } while (0)

}  // namespace

struct ddrl_comm {
    int device, rank, world;
    ncclComm_t comm;
};

extern "C" {

int ddrl_comm_unique_id(uint8_t *id_h) {
    DDRL_REQUIRE(id_h != nullptr, "NULL pointer");
    DDRL_RCCL_READY();
    ncclUniqueId id;
    DDRL_NCCL_CHECK(R.GetUniqueId(&id), "ncclGetUniqueId");
    memcpy(id_h, id.internal, sizeof(id.internal));
    return DDRL_OK;
}

int ddrl_comm_init(ddrl_comm_t **out, int device, int32_t rank, int32_t world, const uint8_t *id_h) {
    DDRL_REQUIRE(out != nullptr && id_h != nullptr, "NULL pointer");
    DDRL_REQUIRE(world >= 1 && rank >= 0 && rank < world, "rank outside [0, world)");
    DDRL_RCCL_READY();
    ddrl::DeviceGuard g(device);
    if (!g.ok) { ddrl::set_error("cannot select device %d", device); return DDRL_ERR_HIP; }
    ncclUniqueId id;
    memcpy(id.internal, id_h, sizeof(id.internal));
    ddrl_comm *h = new ddrl_comm{device, rank, world, nullptr};
    const ncclResult_t rc = R.CommInitRank(&h->comm, world, id, rank);   // collective: every rank of the communicator calls it
    if (rc != 0) { delete h; return nccl_fail("ncclCommInitRank", rc); }
    *out = h;
    return DDRL_OK;
}

int ddrl_comm_destroy(ddrl_comm_t *h) {
    if (!h) return DDRL_OK;
    Rccl &R = rccl();
    ddrl::DeviceGuard g(h->device);
    if (R.lib && h->comm) (void)R.CommDestroy(h->comm);
    delete h;
    return DDRL_OK;
}

int ddrl_comm_bcast_params(ddrl_comm_t *h, float *flat_d, int64_t n, int32_t root, void *stream) {
    DDRL_REQUIRE(h != nullptr && flat_d != nullptr && n > 0 && root >= 0 && root < h->world, "bad handle / buffer / count / root");
    DDRL_RCCL_READY();
    ddrl::DeviceGuard g(h->device);
    DDRL_NCCL_CHECK(R.Broadcast(flat_d, flat_d, (size_t)n, k_ncclFloat32, root, h->comm, ddrl::as_stream(stream)), "ncclBroadcast");
    return DDRL_OK;
}

int ddrl_comm_allreduce_grads(ddrl_comm_t *h, float *flat_d, int64_t n, void *stream) {
    DDRL_REQUIRE(h != nullptr && flat_d != nullptr && n > 0, "bad handle / buffer / count");
    DDRL_RCCL_READY();
    ddrl::DeviceGuard g(h->device);
    DDRL_NCCL_CHECK(R.AllReduce(flat_d, flat_d, (size_t)n, k_ncclFloat32, k_ncclAvg, h->comm, ddrl::as_stream(stream)), "ncclAllReduce");
    return DDRL_OK;
}

int ddrl_comm_send_batch(ddrl_comm_t *h, const float *block_d, int64_t n, int32_t peer, void *stream) {
    DDRL_REQUIRE(h != nullptr && block_d != nullptr && n > 0 && peer >= 0 && peer < h->world, "bad handle / buffer / count / peer");
    DDRL_RCCL_READY();
    ddrl::DeviceGuard g(h->device);
    DDRL_NCCL_CHECK(R.Send(block_d, (size_t)n, k_ncclFloat32, peer, h->comm, ddrl::as_stream(stream)), "ncclSend");
    return DDRL_OK;
}

int ddrl_comm_recv_batch(ddrl_comm_t *h, float *block_d, int64_t n, int32_t peer, void *stream) {
    DDRL_REQUIRE(h != nullptr && block_d != nullptr && n > 0 && peer >= 0 && peer < h->world, "bad handle / buffer / count / peer");
    DDRL_RCCL_READY();
    ddrl::DeviceGuard g(h->device);
    DDRL_NCCL_CHECK(R.Recv(block_d, (size_t)n, k_ncclFloat32, peer, h->comm, ddrl::as_stream(stream)), "ncclRecv");
    return DDRL_OK;
}

int ddrl_comm_group_start(void) {
    DDRL_RCCL_READY();
    DDRL_NCCL_CHECK(R.GroupStart(), "ncclGroupStart");
    return DDRL_OK;
}

int ddrl_comm_group_end(void) {
    DDRL_RCCL_READY();
    DDRL_NCCL_CHECK(R.GroupEnd(), "ncclGroupEnd");
    return DDRL_OK;
}

}  // extern "C"
