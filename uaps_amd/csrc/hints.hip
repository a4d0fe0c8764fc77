#include "hints.hpp"
#include <string.h>
#include <stddef.h>
#include <atomic>

namespace {
thread_local uaps_call_hints g_hints;
thread_local bool g_have = false;
thread_local uaps::LaunchEvents g_launch;
constexpr int kMaxDevices = 64;
unsigned* g_error_word[kMaxDevices] = {};      // one per device: a kernel only ever writes to a word of the device it runs on
// -1: no current device or one beyond the table -- such a device gets NO error word (its kernels report nothing) instead of device
// 0's, which a kernel on another device must never store to
int current_device() { int d = 0; return hipGetDevice(&d) == hipSuccess && d >= 0 && d < kMaxDevices ? d : -1; }
}

namespace uaps {
uaps_call_hints take_hints() {
    uaps_call_hints h;
    if (g_have) { h = g_hints; g_have = false; }
    else memset(&h, 0, sizeof h);
    return h;
}
int read_hints(const uaps_call_hints* in, uaps_call_hints& out) {
    memset(&out, 0, sizeof out);
    if (!in || in->struct_size == 0) return UAPS_OK;
    // size-versioned: a client built against an older, shorter struct is read only as far as ITS struct goes
    const unsigned n = in->struct_size;
    if (n < offsetof(uaps_call_hints, out_amax) || n > sizeof(uaps_call_hints)) return UAPS_EINVAL;
    memcpy(&out, in, n);
    for (int i = 0; i < 3; ++i)
        if (out.bound[i] && !(out.mul[i] > 0.f && out.mul[i] < 3.0e38f)) return UAPS_EINVAL;
    return UAPS_OK;
}
LaunchEvents& launch_events() { return g_launch; }
unsigned* error_word() { const int d = current_device(); return d >= 0 ? g_error_word[d] : nullptr; }
}  // namespace uaps

namespace {
std::atomic<bool> g_acct_on{false};
std::atomic<unsigned long long> g_acct_bytes{0};
}
namespace uaps {
void account_bytes(double bytes) {
    if (g_acct_on.load(std::memory_order_relaxed) && bytes > 0.0) g_acct_bytes.fetch_add((unsigned long long)bytes, std::memory_order_relaxed);
}
}  // namespace uaps
extern "C" int uaps_account(int enable) {                 // 1: reset the tally and count from here; 0: stop counting (the tally stays readable)
    if (enable) g_acct_bytes.store(0);
    g_acct_on.store(enable != 0);
    return UAPS_OK;
}
extern "C" double uaps_accounted_bytes(void) { return (double)g_acct_bytes.load(); }

extern "C" int uaps_set_error_word(unsigned* device_word) {
    if ((uintptr_t)device_word % 4) return UAPS_EINVAL;
    const int d = current_device();
    if (d < 0) return device_word ? UAPS_ERANGE : UAPS_OK;      // (no device: nothing to attach to, nothing to detach)
    g_error_word[d] = device_word;      // the word of the CURRENT device (hipSetDevice), like every launch here; ONE word per device:
    return UAPS_OK;                     // every model / trainer of the process on that device shares it, and whoever reads it clears it for all
}

// Zero fill of bound slots with agent-scope stores: the slots are then raised by memory-side atomics and read with agent-scope
// loads (hints.hpp), so every access to them bypasses the per-XCD L2s.  A plain fill leaves its zeros in one XCD's write-back L2;
// under a hipGraph replay a late write-back of such a line was seen to wipe the raised value (the convolution then scales by a
// different power of two and the replayed step differs from the eager one in the last bits).
namespace {
__global__ void zero_bounds_kernel(float* __restrict__ p, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) __hip_atomic_store(p + i, 0.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
}  // namespace
extern "C" int uaps_zero_bounds(float* p, long n, uaps_stream_t stream) {
    if (!p || n <= 0) return UAPS_EINVAL;
    hipLaunchKernelGGL(zero_bounds_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p, n);
    return (int)hipGetLastError();
}

extern "C" int uaps_next_launch_events(void* start, void* stop) {
    const int used = g_launch.used ? 1 : 0;
    g_launch.start = (hipEvent_t)start; g_launch.stop = (hipEvent_t)stop;
    g_launch.armed = start != nullptr && stop != nullptr;
    g_launch.used = false;
    return used;
}

extern "C" int uaps_next_call_hints(const uaps_call_hints* h) {
    if (!h) { g_have = false; return UAPS_OK; }
    if (h->struct_size == 0) return UAPS_EINVAL;
    uaps_call_hints t;
    const int rc = uaps::read_hints(h, t);
    if (rc) return rc;
    g_hints = t; g_have = true;
    return UAPS_OK;
}
