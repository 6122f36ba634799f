// Error reporting, version and device probes of libddrl_hip.so.
#include "ddrl_common.h"

namespace ddrl {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace ddrl

namespace {
// counter-based noise fills (tf.random_normal / action_space.sample stand-ins)
__global__ void __launch_bounds__(256) k_normal_fill(float *out, long long n, uint32_t seed, unsigned long long counter) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned long long c = counter + (unsigned long long)i;
    const uint32_t lo = (uint32_t)c, hi = (uint32_t)(c >> 32);
    const uint32_t h1 = ddrl::hash3(seed, lo, 2u * hi), h2 = ddrl::hash3(seed, lo, 2u * hi + 1u);
    const float u1 = (float)((h1 >> 8) + 1u) * (1.0f / 16777216.0f);  // (0,1]
    const float u2 = ddrl::u01(h2);
    const float r = sqrtf(-2.0f * logf(u1));
    out[i] = r * cosf(6.28318530717958647692f * u2);
}
// the same fill with the counter read from device memory (two 32-bit words, low first): the launch can sit in a captured graph whose
// replays take their counter from a word the host block carries up (ddrl_sac1_step_host)
__global__ void __launch_bounds__(256) k_normal_fill_ctr(float *out, long long n, uint32_t seed, const uint32_t *__restrict__ ctr, unsigned long long base) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned long long c = ((unsigned long long)ctr[0] | ((unsigned long long)ctr[1] << 32)) + base + (unsigned long long)i;
    const uint32_t lo = (uint32_t)c, hi = (uint32_t)(c >> 32);
    const uint32_t h1 = ddrl::hash3(seed, lo, 2u * hi), h2 = ddrl::hash3(seed, lo, 2u * hi + 1u);
    const float u1 = (float)((h1 >> 8) + 1u) * (1.0f / 16777216.0f);  // (0,1]
    const float u2 = ddrl::u01(h2);
    const float r = sqrtf(-2.0f * logf(u1));
    out[i] = r * cosf(6.28318530717958647692f * u2);
}
// ddrl_sac1_step_host without copy nodes: ONE launch reads the caller's page-locked block straight over PCIe into the learner's input set
// and generates the three noise tensors behind the counter the block carries in its last two words (a captured graph of kernel
// launches only: the two H2D copy nodes of the first form cost ~25 us of device time per replay, more than half an update)
__global__ void __launch_bounds__(256) k_host_block_up(const float *__restrict__ src, float *__restrict__ dst, long long n, float *e0, float *e1, float *e2,
                                                       long long m, uint32_t seed, const uint32_t *__restrict__ ctr, unsigned nb_copy) {
    if (blockIdx.x < nb_copy) {
        const long long i4 = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
        if (i4 + 3 < n) *reinterpret_cast<float4 *>(dst + i4) = *reinterpret_cast<const float4 *>(src + i4);
        else
            for (long long i = i4; i < n; ++i) dst[i] = src[i];
        return;
    }
    const long long i = (long long)(blockIdx.x - nb_copy) * 256 + threadIdx.x;
    if (i >= 3 * m) return;
    const unsigned long long c = ((unsigned long long)ctr[0] | ((unsigned long long)ctr[1] << 32)) + (unsigned long long)i;
    const uint32_t lo = (uint32_t)c, hi = (uint32_t)(c >> 32);
    const uint32_t h1 = ddrl::hash3(seed, lo, 2u * hi), h2 = ddrl::hash3(seed, lo, 2u * hi + 1u);
    const float u1 = (float)((h1 >> 8) + 1u) * (1.0f / 16777216.0f);  // (0,1]
    const float u2 = ddrl::u01(h2);
    const float r = sqrtf(-2.0f * logf(u1));
    float *out = i < m ? e0 : (i < 2 * m ? e1 : e2);
    out[i < m ? i : (i < 2 * m ? i - m : i - 2 * m)] = r * cosf(6.28318530717958647692f * u2);
}
__global__ void __launch_bounds__(256) k_uniform_fill(float *out, long long n, float lo_v, float hi_v, uint32_t seed,
                                                      unsigned long long counter) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned long long c = counter + (unsigned long long)i;
    const uint32_t h = ddrl::hash3(seed, (uint32_t)c, 2u * (uint32_t)(c >> 32));
    out[i] = lo_v + (hi_v - lo_v) * ddrl::u01(h);
}
}  // namespace

// internal (sac1.hip: ddrl_sac1_step_host)
int ddrl_internal_normal_fill_ctr(float *out_d, int64_t n, uint32_t seed, const uint32_t *ctr_d, uint64_t base, void *stream) {
    if (n <= 0) return DDRL_OK;
    k_normal_fill_ctr<<<(unsigned)((n + 255) / 256), 256, 0, ddrl::as_stream(stream)>>>(out_d, n, seed, ctr_d, base);
    DDRL_LAUNCH_CHECK();
    return DDRL_OK;
}

// internal (sac1.hip: ddrl_sac1_step_host): src / ctr are DEVICE-side addresses of page-locked host memory
int ddrl_internal_host_block_up(const float *src, float *dst_d, int64_t n, float *e0, float *e1, float *e2, int64_t m, uint32_t seed,
                                const uint32_t *ctr, void *stream) {
    const unsigned nb_copy = (unsigned)((n + 1023) / 1024), nb_noise = (unsigned)((3 * m + 255) / 256);
    k_host_block_up<<<nb_copy + nb_noise, 256, 0, ddrl::as_stream(stream)>>>(src, dst_d, n, e0, e1, e2, m, seed, ctr, nb_copy);
    DDRL_LAUNCH_CHECK();
    return DDRL_OK;
}

extern "C" {

int ddrl_version(void) { return DDRL_VERSION; }
const char *ddrl_last_error(void) { return ddrl::g_err; }

int ddrl_device_arch(int device, char *buf_h, int buflen) {
    DDRL_REQUIRE(buf_h != nullptr && buflen > 0, "NULL buffer");
    hipDeviceProp_t prop;
    DDRL_HIP_CHECK(hipGetDeviceProperties(&prop, device));
    snprintf(buf_h, (size_t)buflen, "%s", prop.gcnArchName);
    return DDRL_OK;
}

int ddrl_host_device_pointer(const void *host_ptr, void **dev_ptr_out) {
    DDRL_REQUIRE(host_ptr != nullptr && dev_ptr_out != nullptr, "NULL pointer");
    hipPointerAttribute_t pa{};
    if (hipPointerGetAttributes(&pa, host_ptr) != hipSuccess || pa.type != hipMemoryTypeHost || pa.devicePointer == nullptr) {
        (void)hipGetLastError();
        ddrl::set_error("not page-locked host memory the device can address");
        return DDRL_ERR_BAD_ARG;
    }
    *dev_ptr_out = pa.devicePointer;
    return DDRL_OK;
}

int ddrl_normal_fill(float *out_d, int64_t n, uint32_t seed, uint64_t counter, void *stream) {
    DDRL_REQUIRE(n >= 0 && (n == 0 || out_d != nullptr), "NULL output");
    if (n == 0) return DDRL_OK;
    k_normal_fill<<<(unsigned)((n + 255) / 256), 256, 0, ddrl::as_stream(stream)>>>(out_d, n, seed, counter);
    DDRL_LAUNCH_CHECK();
    return DDRL_OK;
}

int ddrl_uniform_fill(float *out_d, int64_t n, float lo, float hi, uint32_t seed, uint64_t counter, void *stream) {
    DDRL_REQUIRE(n >= 0 && (n == 0 || out_d != nullptr), "NULL output");
    if (n == 0) return DDRL_OK;
    k_uniform_fill<<<(unsigned)((n + 255) / 256), 256, 0, ddrl::as_stream(stream)>>>(out_d, n, lo, hi, seed, counter);
    DDRL_LAUNCH_CHECK();
    return DDRL_OK;
}

}  // extern "C"
