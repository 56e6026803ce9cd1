// common.h -- shared device/host helpers for libgrafp_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "grafp_hip.h"

namespace grafp {

void set_error(const char *fmt, ...);

// MI355X: 8 XCDs with private L2s; block b is dispatched to XCD b % 8 (observed, speed only).
// Remap so that each XCD owns one CONTIGUOUS range of logical tiles: tiles that share operand
// panels (all query tiles of one clip, all query groups over one database slice) then hit the same L2.
// Bijective for every n (cdna_hip_programming.md, "XCD swizzle must be bijective").
__device__ __forceinline__ int xcd_remap(int bid, int n) {
    const int xcd = bid & 7, slot = bid >> 3;
    const int q = n >> 3, r = n & 7;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + slot;
}

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// exact-f32 MFMA: D(32x32) += A(32x2) * B(2x32); lane l supplies A[l&31][l>>5] and B[l>>5][l&31];
// D[row][col]: col = l&31, row = (reg&3) + 8*(reg>>2) + 4*(l>>5).  Bitwise a k-ordered fmaf chain.
__device__ __forceinline__ f32x16 mfma32x32x2(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ int mfma_row(int reg, int half) { return (reg & 3) + 8 * (reg >> 2) + 4 * half; }

}  // namespace grafp

// Streaming (non-temporal) store of a result row piece: the consumer is a later launch and the producers' working set
// should keep the caches.  Experiment builds (make measure XFLAGS=-DGRAFP_PLAIN_STORES=f MLIB=... MDIR=...) turn them
// into plain stores for the A/B of tools/step_lib_ab.py at Infinity-Cache-resident tensor sizes: f = 9 everywhere, or
// only in the family whose source defines GRAFP_STORE_FAMILY = f (1 BatchNorm, 2 products, 3 max-relative).
#ifndef GRAFP_STORE_FAMILY
#define GRAFP_STORE_FAMILY 0
#endif
#if defined(GRAFP_PLAIN_STORES) && (GRAFP_PLAIN_STORES == 9 || GRAFP_PLAIN_STORES == GRAFP_STORE_FAMILY)
#define GRAFP_ST_NT(v, p) (*(p) = (v))
#else
#define GRAFP_ST_NT(v, p) __builtin_nontemporal_store(v, p)
#endif

// A result store whose hint is chosen at RUN time (a wave-uniform flag from the launch plan): `plain` = an ordinary store
// (the tensor fits the Infinity Cache and its readers are the next launches), otherwise the streaming store of
// GRAFP_ST_NT.  Issued as inline asm: written as `if (plain) *p = v; else __builtin_nontemporal_store(v, p);` hipcc 7.2
// sinks the two arms into ONE store and drops the hint (seen in the ISA: no `nt` left in the kernel).
namespace grafp {
typedef unsigned st_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned st_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void store16_hint(void *p, st_u32x4 v, bool plain) {
    if (plain) asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void store8_hint(void *p, st_u32x2 v, bool plain) {
    if (plain) asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx2 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
}
}  // namespace grafp

// A load of data that is read ONCE (a saved activation in backward, a product's raw output in its normalise pass): the
// non-temporal hint keeps it from displacing what the next launches re-read.  GRAFP_NT_LOADS is a bit mask of the sites
// that use it (experiment builds override it: make measure XFLAGS=-DGRAFP_NT_LOADS=n):
//   1 bn_bwd1: x   2 bn_affine: y   4 bn_bwd1: dz   8 bn_affine: shortcut   16 mrconv_bwd: g   32 mrconv_fwd: x
// Whole-step A/B of builds (tools/step_lib_ab.py, graph replays; profiles/r06_l_nt_loads_*.txt), 128 / 256 / 1024 pairs:
// mask 1: -1.0 / -0.8 / -0.4 %;  2: -0.2 / -0.4 / -0.8 %;  7: -1.1 / -1.0 / -1.0 ... -1.4 %;  15: -1.3 / -1.0 / -1.5 %;
// 63: -1.3 / -1.7 / -1.4 %  -> all six sites.  (The products' X operand is NOT such a load: its column panel is re-read by
// the other row tiles through L2, and the hint cost 3 % there in round 2.)
#ifndef GRAFP_NT_LOADS
#define GRAFP_NT_LOADS 63
#endif
namespace grafp {
template <typename V> __device__ __forceinline__ V ld_once_nt(const V *p) {
    if constexpr (sizeof(V) == 16) {
        typedef unsigned e4 __attribute__((ext_vector_type(4)));
        return __builtin_bit_cast(V, __builtin_nontemporal_load(reinterpret_cast<const e4 *>(p)));
    } else if constexpr (sizeof(V) == 8) {
        typedef unsigned e2 __attribute__((ext_vector_type(2)));
        return __builtin_bit_cast(V, __builtin_nontemporal_load(reinterpret_cast<const e2 *>(p)));
    } else {
        return *p;
    }
}
}  // namespace grafp
#define GRAFP_LD_ONCE(bit, p) (((GRAFP_NT_LOADS) & (bit)) ? grafp::ld_once_nt(p) : *(p))

#define GRAFP_REQUIRE(cond, ...)              \
    do {                                      \
        if (!(cond)) {                        \
            grafp::set_error(__VA_ARGS__);    \
            return GRAFP_ERR_ARG;             \
        }                                     \
    } while (0)

#define GRAFP_CHECK_LAUNCH(what)                                                      \
    do {                                                                              \
        hipError_t e__ = hipGetLastError();                                           \
        if (e__ != hipSuccess) {                                                      \
            grafp::set_error("%s: launch failed: %s", what, hipGetErrorString(e__));  \
            return GRAFP_ERR_LAUNCH;                                                  \
        }                                                                             \
    } while (0)
