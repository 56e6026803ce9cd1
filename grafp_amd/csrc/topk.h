// topk.h -- (distance, id) top-k selection pieces shared by knn_search.hip and ivfpq.hip (gfx950, 64-lane waves).
#pragma once
#include <math.h>

#include "common.h"

namespace grafp {

constexpr int SR_EMPTY = 0x7fffffff;

// Order LDS traffic between the lanes of ONE wave: wait for this wave's LDS operations only (lgkmcnt) -- a full
// fence would also drain vmcnt, i.e. stall on the database prefetch that is deliberately left in flight.
#define WAVE_SYNC()                                                   \
    do {                                                              \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            \
        __builtin_amdgcn_wave_barrier();                              \
    } while (0)

template <typename I>
__device__ __forceinline__ bool lex_lt(float d1, I i1, float d2, I i2) {
    return d1 < d2 || (d1 == d2 && i1 < i2);
}

// value of lane (lane ^ j), j a power of two (compile-time after unrolling): DPP for 1, 2, 8 (no LDS crossbar trip),
// ds_swizzle for 4 and 16, ds_bpermute only across the two 32-lane halves
__device__ __forceinline__ int xor_lane(int v, int j) {
    switch (j) {
        case 1: return __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);    // quad_perm [1,0,3,2]
        case 2: return __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);    // quad_perm [2,3,0,1]
        case 8: return __builtin_amdgcn_update_dpp(0, v, 0x128, 0xF, 0xF, true);   // row_ror:8
        case 4: return __builtin_amdgcn_ds_swizzle(v, (4 << 10) | 0x1F);           // bit mode: xor 4
        case 16: return __builtin_amdgcn_ds_swizzle(v, (16 << 10) | 0x1F);         // bit mode: xor 16
        default: return __shfl_xor(v, j);
    }
}
__device__ __forceinline__ float xor_lane(float v, int j) { return __int_as_float(xor_lane(__float_as_int(v), j)); }
__device__ __forceinline__ long long xor_lane(long long v, int j) {
    const int lo = xor_lane((int)(v & 0xffffffffll), j), hi = xor_lane((int)(v >> 32), j);
    return ((long long)hi << 32) | (unsigned int)lo;
}

// 64-lane bitonic sort, one (d, i) pair per lane, ascending by (d, i)
template <typename I>
__device__ __forceinline__ void wave_sort64(float &d, I &i, int lane) {
#pragma unroll
    for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            const float od = xor_lane(d, j);
            const I oi = xor_lane(i, j);
            const bool want_min = ((lane & j) == 0) == ((lane & k) == 0);
            const bool take = want_min ? lex_lt(od, oi, d, i) : lex_lt(d, i, od, oi);
            d = take ? od : d;
            i = take ? oi : i;
        }
    }
}

// Running top-32 of one wave: sorted list in lanes 0..31 of (td, ti); survivors of the threshold test are compacted
// (ballot prefix, no atomics) into an LDS queue and folded in 32 at a time.  thr only ever tightens.
constexpr int WT_PEND = 96;      // <= 31 left over + 64 new per push
struct WaveTop {
    float td;
    int ti;
    float thr;
    int pc;
    bool empty;
    float *pd;
    int *pi;
    __device__ __forceinline__ void init(float *qd, int *qi, float thr0) {
        empty = true;
        td = INFINITY;
        ti = SR_EMPTY;
        thr = thr0;
        pc = 0;
        pd = qd;
        pi = qi;
    }
    __device__ __forceinline__ void fold(int k, int lane) {
        WAVE_SYNC();
        int off = 0;
        if (empty) {                       // nothing kept yet: the first sort takes 64 queue entries
            td = lane < pc ? pd[lane] : INFINITY;
            ti = lane < pc ? pi[lane] : SR_EMPTY;
            wave_sort64(td, ti, lane);
            off = 64;
            empty = false;
        }
        for (; off < pc; off += 32) {
            float d = td;
            int i = ti;
            if (lane >= 32) {
                const int e = off + lane - 32;
                d = e < pc ? pd[e] : INFINITY;
                i = e < pc ? pi[e] : SR_EMPTY;
            }
            wave_sort64(d, i, lane);
            td = d;                        // lanes 0..31: the 32 best so far
            ti = i;
        }
        thr = fminf(thr, __shfl(td, k - 1));
        pc = 0;
        WAVE_SYNC();
    }
    // wave-uniform call; every lane offers one (d, i) or nothing
    __device__ __forceinline__ void push(bool valid, float d, int i, int k, int lane) {
        const bool pass = valid && d <= thr;
        const unsigned long long m = __ballot(pass);
        if (m == 0) return;
        if (pass) {
            const int pos = pc + __popcll(m & ((1ull << lane) - 1ull));
            pd[pos] = d;
            pi[pos] = i;
        }
        pc += __popcll(m);
        if (pc >= 32) fold(k, lane);
    }
};

__device__ __forceinline__ void block_merge_tops(WaveTop &top, float (*wtop_d)[32], int (*wtop_i)[32], int wave, int lane,
                                                 float &td, int &ti) {
    // merge the four wave lists as a tree: (0,1) and (2,3) in parallel, then the two winners -> wave 0 lanes 0..31
    if ((wave & 1) && lane < 32) {
        wtop_d[wave][lane] = top.td;
        wtop_i[wave][lane] = top.ti;
    }
    __syncthreads();
    td = top.td;
    ti = top.ti;
    if (!(wave & 1)) {
        if (lane >= 32) {
            td = wtop_d[wave + 1][lane - 32];
            ti = wtop_i[wave + 1][lane - 32];
        }
        wave_sort64(td, ti, lane);
        if (wave == 2 && lane < 32) {
            wtop_d[2][lane] = td;
            wtop_i[2][lane] = ti;
        }
    }
    __syncthreads();
    if (wave == 0) {
        if (lane >= 32) {
            td = wtop_d[2][lane - 32];
            ti = wtop_i[2][lane - 32];
        }
        wave_sort64(td, ti, lane);
    }
}

}  // namespace grafp
