// Parameter server buffer: one flat float32 device array, push = snapshot-by-copy,
// pull = copy-out.  Replaces class ParameterServer (example/dsac.py:51-73); the key -> (offset,
// shape) table is kept by the Python ParameterServer class.
#include "ddrl_common.h"

struct ddrl_ps {
    int device;
    float *buf;
    long long count;
    long long version;
};

extern "C" {

int ddrl_ps_create(ddrl_ps_t **out, int device, int64_t count) {
    DDRL_REQUIRE(out != nullptr && count > 0, "bad out/count");
    ddrl::DeviceGuard g(device);
    if (!g.ok) { ddrl::set_error("cannot select device %d", device); return DDRL_ERR_HIP; }
    ddrl_ps *h = new ddrl_ps();
    h->device = device; h->count = count; h->version = 0; h->buf = nullptr;
    if (hipMalloc(&h->buf, (size_t)count * sizeof(float)) != hipSuccess) {
        ddrl::set_error("hipMalloc failed for parameter server of %lld floats", (long long)count);
        delete h;
        return DDRL_ERR_NOMEM;
    }
    DDRL_HIP_CHECK(hipMemset(h->buf, 0, (size_t)count * sizeof(float)));
    *out = h;
    return DDRL_OK;
}

int ddrl_ps_destroy(ddrl_ps_t *h) {
    if (!h) return DDRL_OK;
    ddrl::DeviceGuard g(h->device);
    (void)hipFree(h->buf);
    delete h;
    return DDRL_OK;
}

int ddrl_ps_push(ddrl_ps_t *h, const float *src_d, int64_t offset, int64_t count, void *stream) {
    DDRL_REQUIRE(h != nullptr && src_d != nullptr, "NULL pointer");
    DDRL_REQUIRE(offset >= 0 && count >= 0 && offset + count <= h->count, "range outside the server buffer");
    ddrl::DeviceGuard g(h->device);
    if (count)
        DDRL_HIP_CHECK(hipMemcpyAsync(h->buf + offset, src_d, (size_t)count * sizeof(float), hipMemcpyDeviceToDevice,
                                      ddrl::as_stream(stream)));
    h->version += 1;
    return DDRL_OK;
}

int ddrl_ps_pull(ddrl_ps_t *h, float *dst_d, int64_t offset, int64_t count, void *stream) {
    DDRL_REQUIRE(h != nullptr && dst_d != nullptr, "NULL pointer");
    DDRL_REQUIRE(offset >= 0 && count >= 0 && offset + count <= h->count, "range outside the server buffer");
    ddrl::DeviceGuard g(h->device);
    if (count)
        DDRL_HIP_CHECK(hipMemcpyAsync(dst_d, h->buf + offset, (size_t)count * sizeof(float), hipMemcpyDeviceToDevice,
                                      ddrl::as_stream(stream)));
    return DDRL_OK;
}

int ddrl_ps_buffer(ddrl_ps_t *h, float **buf_d, int64_t *count) {
    DDRL_REQUIRE(h != nullptr, "handle is NULL");
    if (buf_d) *buf_d = h->buf;
    if (count) *count = h->count;
    return DDRL_OK;
}

int64_t ddrl_ps_version(ddrl_ps_t *h) { return h ? h->version : -1; }

}  // extern "C"
