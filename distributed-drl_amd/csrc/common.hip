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
