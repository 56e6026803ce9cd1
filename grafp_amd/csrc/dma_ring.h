// dma_ring.h -- pieces shared by the LDS-DMA ring kernels (gemm.hip, wgrad.hip), gfx950 only.
#pragma once
#include <type_traits>

#include "common.h"

namespace grafp {

typedef short gm_bf16x8 __attribute__((ext_vector_type(8)));
typedef short gm_s16x4 __attribute__((ext_vector_type(4)));
typedef float gm_f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned gm_u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 gm_bf16x2 __attribute__((ext_vector_type(2)));
typedef const void __attribute__((address_space(1))) *gm_gptr;
typedef void __attribute__((address_space(3))) *gm_lptr;

__device__ __forceinline__ unsigned gm_pack_bf16(float a, float b) {
    const gm_f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, gm_bf16x2));   // v_cvt_pk_bf16_f32 (RNE)
}

// One LDS-DMA instruction: 64 lanes x 16 bytes from per-lane global addresses to LDS [lds_base + lane * 16).  Issued
// through inline asm ON PURPOSE: hipcc's wait-count pass treats every ds_read after a builtin LDS-DMA as a possible
// reader of its destination and drains vmcnt(0) in front of it (seen in the .s: one full drain per chunk), which
// would serialise the ring.  Here the DMA is invisible to that pass and the waits are counted by hand (below).
__device__ __forceinline__ void gm_dma16(const void *gsrc, unsigned lds_base) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(lds_base) : "memory", "m0");
}

// the 4-bytes-per-lane form (256 bytes per instruction): small vectors that must not pass through the compiler's own
// wait counting while 16-byte DMAs are in flight
__device__ __forceinline__ void gm_dma4(const void *gsrc, unsigned lds_base) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" ::"v"(gsrc), "s"(lds_base) : "memory", "m0");
}

template <int N> __device__ __forceinline__ void gm_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// at most `allowed` of this wave's newest vector-memory operations may still be in flight (rounded DOWN to a step)
__device__ __forceinline__ void gm_wait_allowed(int allowed) {
    if (allowed >= 24) {
        if (allowed >= 48) gm_wait_vm<48>();
        else if (allowed >= 40) gm_wait_vm<40>();
        else if (allowed >= 36) gm_wait_vm<36>();
        else if (allowed >= 32) gm_wait_vm<32>();
        else gm_wait_vm<24>();
    } else if (allowed >= 12) {
        if (allowed >= 20) gm_wait_vm<20>();
        else if (allowed >= 16) gm_wait_vm<16>();
        else gm_wait_vm<12>();
    } else {
        if (allowed >= 8) gm_wait_vm<8>();
        else if (allowed >= 4) gm_wait_vm<4>();
        else gm_wait_vm<0>();
    }
}

}  // namespace grafp
