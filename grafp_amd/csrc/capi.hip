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
