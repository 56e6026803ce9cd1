// capi.hip -- ABI version and thread-local error message.
#include <stdarg.h>
#include <stdio.h>

#include "common.h"

namespace grafp {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace grafp

extern "C" int grafp_abi_version(void) { return GRAFP_ABI_VERSION; }
extern "C" const char *grafp_last_error(void) { return grafp::g_err; }

// Test utility: `blocks` workgroups of `threads` threads that do nothing but hold their CU slots for `clocks` shader
// cycles (tests/test_gpu_kernels.py runs the BatchNorm rendezvous next to it -- the stand-in for a collective's kernels
// occupying CUs while backward runs).
namespace grafp {
__global__ void occupy_kernel(long long clocks, int *sink) {
    const long long t0 = __builtin_readcyclecounter();
    int spins = 0;
    while ((long long)__builtin_readcyclecounter() - t0 < clocks) {
        __builtin_amdgcn_s_sleep(32);
        ++spins;
    }
    if (sink && spins < 0) *sink = spins;
}
}  // namespace grafp
extern "C" int grafp_debug_occupy(int blocks, int threads, int64_t clocks, grafp_stream_t stream) {
    GRAFP_REQUIRE(blocks > 0 && threads > 0 && threads <= 1024 && clocks >= 0, "debug_occupy: bad arguments");
    hipLaunchKernelGGL(grafp::occupy_kernel, dim3(blocks), dim3(threads), 0, (hipStream_t)stream, (long long)clocks,
                       (int *)nullptr);
    GRAFP_CHECK_LAUNCH("occupy_kernel");
    return GRAFP_OK;
}

// Progress flags: how a stream OUTSIDE a replayed HIP graph learns that the graph has passed a certain point.
// (hipEventRecordExternal -- the event-record node made for this -- returns "invalid argument" under stream capture on
// the HIP 7.0 runtime this image ships, and PyTorch refuses external events on ROCm for that reason.)  The graph contains
// grafp_flag_bump launches (flag += 1, release, device scope) behind the work they mark; the other stream runs
// grafp_flag_wait (one wave polling with s_sleep until flag >= value, acquire).  The data-parallel step replays its
// backward graph and starts each gradient bucket's all-reduce on a communication stream as soon as the graph has packed
// the bucket (grafp_amd/dist.py: GradSync.begin_capture).  The wait is enqueued AFTER the graph launch and is bounded:
// after ~10 s of polling it traps instead of hanging the device.
namespace grafp {
__global__ void flag_bump_kernel(int *flag) {
    if (threadIdx.x == 0) {
        __threadfence();
        __hip_atomic_fetch_add(flag, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
}
__global__ void flag_wait_kernel(const int *flag, int value) {
    if (threadIdx.x == 0) {
        const long long t0 = __builtin_readcyclecounter();
        while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - value < 0) {
            __builtin_amdgcn_s_sleep(64);
            if ((long long)__builtin_readcyclecounter() - t0 > 2000000000ll) __builtin_trap();      // 1 s at 2 GHz, 20 s at 100 MHz
        }
    }
}
}  // namespace grafp
extern "C" int grafp_flag_bump(int32_t *flag, grafp_stream_t stream) {
    GRAFP_REQUIRE(flag, "flag_bump: null pointer");
    hipLaunchKernelGGL(grafp::flag_bump_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (int *)flag);
    GRAFP_CHECK_LAUNCH("flag_bump_kernel");
    return GRAFP_OK;
}
extern "C" int grafp_flag_wait(const int32_t *flag, int32_t value, grafp_stream_t stream) {
    GRAFP_REQUIRE(flag, "flag_wait: null pointer");
    hipLaunchKernelGGL(grafp::flag_wait_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const int *)flag, (int)value);
    GRAFP_CHECK_LAUNCH("flag_wait_kernel");
    return GRAFP_OK;
}
