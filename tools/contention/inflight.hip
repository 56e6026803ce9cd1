// inflight.hip -- the smallest victim: a copy kernel that keeps N 16-byte loads per thread in flight (N x 4 VGPRs live until
// the last one has returned) and then writes them back.  No LDS, no barrier, no inline asm, no dependence between
// workgroups; plain hipcc output.  Used by two_stream.py (victim "copyN").
#include <hip/hip_runtime.h>
#include <stdint.h>

template <int N>
__global__ __launch_bounds__(256) void inflight_copy(const float4 *__restrict__ in, float4 *__restrict__ out, int64_t n) {
    const int64_t base = (int64_t)blockIdx.x * 256 * N + threadIdx.x;
    float4 r[N];
#pragma unroll
    for (int i = 0; i < N; ++i) r[i] = in[base + (int64_t)i * 256];
    __builtin_amdgcn_sched_barrier(0);            // every load is issued before the first store (else hipcc interleaves them)
#pragma unroll
    for (int i = 0; i < N; ++i) out[base + (int64_t)i * 256] = r[i];
}

// A pure arithmetic victim: every thread runs a chain of `iters` x 8 multiply-adds on 16 registers and stores the result --
// PK: on float2 operands (hipcc emits v_pk_fma_f32), otherwise on scalars (v_fma_f32).  The result is a deterministic
// function of (thread, iters): any difference between two launches is a wrong ALU result or a wrong register.
typedef float f2 __attribute__((ext_vector_type(2)));
template <bool PK>
__global__ __launch_bounds__(256) void alu_chain(const float *__restrict__ in, float *__restrict__ out, int iters) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (PK) {
        f2 a[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = f2{in[(t * 16 + 2 * i) & 65535], in[(t * 16 + 2 * i + 1) & 65535]};
        const f2 m = f2{0.999f, 1.001f}, c = f2{0.001f, -0.001f};
        for (int k = 0; k < iters; ++k) {
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] = __builtin_elementwise_fma(a[i], m, c + a[(i + 1) & 7] * 1e-3f);
        }
        f2 s = a[0];
#pragma unroll
        for (int i = 1; i < 8; ++i) s += a[i];
        out[t] = s.x + s.y;
    } else {
        float a[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) a[i] = in[(t * 16 + i) & 65535];
        for (int k = 0; k < iters; ++k) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float m = (i & 1) ? 1.001f : 0.999f, c = (i & 1) ? -0.001f : 0.001f;
                a[i] = __builtin_fmaf(a[i], m, c + a[(i + 2) & 15] * 1e-3f);
            }
        }
        float s = a[0];
#pragma unroll
        for (int i = 1; i < 16; ++i) s += a[i];
        out[t] = s;
    }
}

// An LDS victim: every thread writes 8 words of a pattern into the workgroup's LDS, then for `rounds` rounds sleeps and
// reads them back; out[block] = number of words that came back different (0 in a correct machine, whatever else runs).
__global__ __launch_bounds__(256) void lds_hold(int *__restrict__ out, int rounds, int lds_words) {
    extern __shared__ unsigned lds[];
    const int tid = threadIdx.x;
    for (int i = tid; i < lds_words; i += 256) lds[i] = 0x9e3779b9u * (unsigned)(i + 1) + blockIdx.x;
    __syncthreads();
    int bad = 0;
    for (int r = 0; r < rounds; ++r) {
        __builtin_amdgcn_s_sleep(32);
        for (int i = tid; i < lds_words; i += 256) bad += lds[i] != 0x9e3779b9u * (unsigned)(i + 1) + blockIdx.x;
    }
    for (int o = 32; o > 0; o >>= 1) bad += __shfl_xor(bad, o);
    if ((tid & 63) == 0 && bad) atomicAdd(out + blockIdx.x, bad);
}

// LDS victims with BANK CONFLICTS, the way the log-mel kernel has them (lane l and lane l + 32 of a wave address the same
// bank at different rows -- its two half-waves own buffers 8 448 bytes apart) and the way a transposition has them:
//   mode 0: word (tid & 31) + (tid >> 5) * 2112            2-way conflict between the halves of every wave
//   mode 1: word tid * 64 (mod the buffer)                  every lane on one bank
//   mode 2: no LDS; __shfl_xor butterflies (ds_bpermute / DPP) over values that are a function of the lane
// Every round rewrites and re-reads; out[block] counts words that came back different from what the thread wrote.
__global__ __launch_bounds__(256) void lds_conflict(int *__restrict__ out, int rounds, int mode) {
    __shared__ unsigned buf[8 * 2112];
    const int tid = threadIdx.x;
    int bad = 0;
    if (mode == 2) {
        for (int r = 0; r < rounds * 8; ++r) {
            unsigned v = (unsigned)(tid & 63) * 2654435761u + (unsigned)r, want = 0;
            for (int l = 0; l < 64; ++l) want ^= (unsigned)l * 2654435761u + (unsigned)r;
            for (int o = 32; o > 0; o >>= 1) v ^= (unsigned)__shfl_xor((int)v, o);
            bad += v != want;
        }
    } else {
        const int idx = mode == 0 ? (tid & 31) + (tid >> 5) * 2112 : (tid * 64) % (8 * 2112);
        for (int r = 0; r < rounds; ++r) {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int at = mode == 0 ? idx + 33 * k : (idx + k) % (8 * 2112);
                buf[at] = (unsigned)(r * 16 + k) * 2654435761u + (unsigned)tid;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int at = mode == 0 ? idx + 33 * k : (idx + k) % (8 * 2112);
                bad += buf[at] != (unsigned)(r * 16 + k) * 2654435761u + (unsigned)tid;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
    for (int o = 32; o > 0; o >>= 1) bad += __shfl_xor(bad, o);
    if ((tid & 63) == 0 && bad) atomicAdd(out + blockIdx.x, bad);
}

extern "C" int lds_conflict_launch(void *out, int blocks, int rounds, int mode, void *stream) {
    hipLaunchKernelGGL(lds_conflict, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (int *)out, rounds, mode);
    return hipGetLastError() == hipSuccess ? 0 : 3;
}

extern "C" int lds_hold_launch(void *out, int blocks, int rounds, int lds_bytes, void *stream) {
    (void)hipFuncSetAttribute((const void *)lds_hold, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    hipLaunchKernelGGL(lds_hold, dim3(blocks), dim3(256), lds_bytes, (hipStream_t)stream, (int *)out, rounds, lds_bytes / 4);
    return hipGetLastError() == hipSuccess ? 0 : 3;
}

// A barrier victim: in every round each thread publishes a round-dependent word in LDS, the workgroup meets at a barrier,
// and each thread reads the word of a thread in ANOTHER wave; a second barrier closes the round.  A read that sees the
// previous round's word means a barrier let a wave through early.  out[block] = number of such reads.
__global__ __launch_bounds__(256) void barrier_ring(int *__restrict__ out, int rounds, int work) {
    __shared__ unsigned slot[256];
    const int tid = threadIdx.x;
    int bad = 0;
    float x = (float)tid;
    for (int r = 0; r < rounds; ++r) {
        for (int k = 0; k < work * (1 + (tid >> 6)); ++k) x = __builtin_fmaf(x, 0.999f, 0.5f);     // waves arrive at different times
        slot[tid] = (unsigned)r * 256u + (unsigned)tid + (x == 12345.0f);
        __syncthreads();
        const int peer = (tid + 64) & 255;
        bad += slot[peer] != (unsigned)r * 256u + (unsigned)peer;
        __syncthreads();
    }
    for (int o = 32; o > 0; o >>= 1) bad += __shfl_xor(bad, o);
    if ((tid & 63) == 0 && bad) atomicAdd(out + blockIdx.x, bad);
}

extern "C" int barrier_ring_launch(void *out, int blocks, int rounds, int work, void *stream) {
    hipLaunchKernelGGL(barrier_ring, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (int *)out, rounds, work);
    return hipGetLastError() == hipSuccess ? 0 : 3;
}

// The arithmetic victim again, with the multiplier where the log-mel and BatchNorm kernels have theirs: SCALAR -- a uniform
// value in an SGPR, read by the vector instruction through the constant bus (v_fma_f32 v, s, v, v) -- or, as the control,
// the same value in a VGPR.  out[thread] is a deterministic function of (thread, iters, k).
template <bool SCALAR>
__global__ __launch_bounds__(256) void sgpr_chain(const float *__restrict__ in, float *__restrict__ out, int iters, float k0,
                                                  float k1) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    float a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = in[(t * 8 + i) & 65535];
    float vk0 = k0, vk1 = k1;
    asm volatile("" : "+v"(vk0), "+v"(vk1));                  // the control's copies live in VGPRs
    for (int r = 0; r < iters; ++r) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (SCALAR) asm volatile("v_fma_f32 %0, %1, %0, %2\n\tv_fma_f32 %0, %3, %0, %2" : "+v"(a[i]) : "s"(k0), "v"(a[(i + 1) & 7] * 1e-3f), "s"(k1));
            else asm volatile("v_fma_f32 %0, %1, %0, %2\n\tv_fma_f32 %0, %3, %0, %2" : "+v"(a[i]) : "v"(vk0), "v"(a[(i + 1) & 7] * 1e-3f), "v"(vk1));
        }
    }
    float s = a[0];
#pragma unroll
    for (int i = 1; i < 8; ++i) s += a[i];
    out[t] = s;
}

extern "C" int sgpr_chain_launch(const void *in, void *out, int blocks, int iters, int scalar, void *stream) {
    if (scalar) hipLaunchKernelGGL(sgpr_chain<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float *)in, (float *)out, iters, 0.999f, 1.001f);
    else hipLaunchKernelGGL(sgpr_chain<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float *)in, (float *)out, iters, 0.999f, 1.001f);
    return hipGetLastError() == hipSuccess ? 0 : 3;
}

// ... and with MANY live registers: NREG accumulators per thread (NREG + ~10 VGPRs), each updated in turn by plain hipcc
// arithmetic (no inline asm) from its neighbour, so every one of them is read and written in every round.
template <int NREG>
__global__ __launch_bounds__(256) void wide_chain(const float *__restrict__ in, float *__restrict__ out, int iters, float k0) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    float a[NREG];
#pragma unroll
    for (int i = 0; i < NREG; ++i) a[i] = in[(t * 8 + i) & 65535];
    for (int r = 0; r < iters; ++r) {
#pragma unroll
        for (int i = 0; i < NREG; ++i) a[i] = __builtin_fmaf(a[i], k0, a[(i + 1) % NREG] * 1e-3f);
    }
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < NREG; ++i) s += a[i];
    out[t] = s;
}

extern "C" int wide_chain_launch(const void *in, void *out, int blocks, int iters, int nreg, void *stream) {
    if (nreg == 64) hipLaunchKernelGGL(wide_chain<64>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float *)in, (float *)out, iters, 0.999f);
    else if (nreg == 128) hipLaunchKernelGGL(wide_chain<128>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float *)in, (float *)out, iters, 0.999f);
    else if (nreg == 224) hipLaunchKernelGGL(wide_chain<224>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float *)in, (float *)out, iters, 0.999f);
    else return 2;
    return hipGetLastError() == hipSuccess ? 0 : 3;
}

// The packed-f32 forms hipcc emits for complex arithmetic: the second source taken with a lane SELECT (op_sel / op_sel_hi:
// "the low result lane reads the pair's high register" and vice versa).  SEL: chains of
//   v_pk_mul_f32 d, d, k op_sel_hi:[1,0]      and      v_pk_add_f32 d, d, c op_sel:[0,1] op_sel_hi:[1,0]
// otherwise the same chains without the selects (MODE below).  out[thread] is a deterministic function of (thread, iters).
template <int MODE>
__global__ __launch_bounds__(256) void pk_sel_chain(const float *__restrict__ in, float *__restrict__ out, int iters) {
    // MODE 0: no selects; 1: both forms below; 2: only  v_pk_mul_f32 d, d, k op_sel_hi:[1,0]  (broadcast of the LOW register);
    // 3: only  v_pk_add_f32 d, d, c op_sel:[0,1] op_sel_hi:[1,0]  (the pair swapped); 4: only  v_pk_add_f32 d, d, c op_sel:[0,1]
    // (broadcast of the HIGH register); 5: v_pk_fma_f32 d, d, k, c op_sel_hi:[1,0,1]  (the form in this repository's scan kernel);
    // 6 / 7: v_pk_fma_f32 with the high-register select on the first / the third source
    const int t = blockIdx.x * 256 + threadIdx.x;
    f2 a[8], k = f2{0.9995f, 1.0005f}, c[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        a[i] = f2{in[(t * 16 + 2 * i) & 65535], in[(t * 16 + 2 * i + 1) & 65535]};
        c[i] = f2{1e-3f * (float)(i + 1), -1e-3f * (float)(i + 2)};
    }
    for (int r = 0; r < iters; ++r) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 5) {
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[1,0,1]" : "+v"(a[i]) : "v"(k), "v"(c[i]));
                continue;
            }
            if (MODE == 6) {                                  // the select on the FIRST source (the form in peak_fwd8)
                asm volatile("v_pk_fma_f32 %0, %1, %0, %2 op_sel:[1,0,0]" : "+v"(a[i]) : "v"(k), "v"(c[i]));
                continue;
            }
            if (MODE == 8) {                                  // ... on the SECOND source of an fma (the form in peak_bwd8)
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel:[0,1,0]" : "+v"(a[i]) : "v"(k), "v"(c[i]));
                continue;
            }
            if (MODE == 9) {                                  // ... on the second source of a multiply, then a plain add
                asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel:[0,1]" : "+v"(a[i]) : "v"(k));
                asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c[i]));
                continue;
            }
            if (MODE == 7) {                                  // the select on the THIRD source
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel:[0,0,1]" : "+v"(a[i]) : "v"(k), "v"(c[i]));
                continue;
            }
            if (MODE == 1 || MODE == 2) asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel_hi:[1,0]" : "+v"(a[i]) : "v"(k));
            else asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(k));
            if (MODE == 1 || MODE == 3) asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0]" : "+v"(a[i]) : "v"(c[i]));
            else if (MODE == 4) asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1]" : "+v"(a[i]) : "v"(c[i]));
            else asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c[i]));
        }
    }
    f2 s2 = a[0];
#pragma unroll
    for (int i = 1; i < 8; ++i) s2 += a[i];
    out[t] = s2.x + 3.0f * s2.y;
}

extern "C" int pk_sel_chain_launch(const void *in, void *out, int blocks, int iters, int mode, void *stream) {
#define PKL(M) hipLaunchKernelGGL(pk_sel_chain<M>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float *)in, (float *)out, iters)
    switch (mode) {
        case 0: PKL(0); break;
        case 1: PKL(1); break;
        case 2: PKL(2); break;
        case 3: PKL(3); break;
        case 4: PKL(4); break;
        case 5: PKL(5); break;
        case 6: PKL(6); break;
        case 7: PKL(7); break;
        case 8: PKL(8); break;
        case 9: PKL(9); break;
        default: return 2;
    }
#undef PKL
    return hipGetLastError() == hipSuccess ? 0 : 3;
}

extern "C" int alu_chain_launch(const void *in, void *out, int blocks, int iters, int packed, void *stream) {
    if (packed) hipLaunchKernelGGL(alu_chain<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float *)in, (float *)out, iters);
    else hipLaunchKernelGGL(alu_chain<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float *)in, (float *)out, iters);
    return hipGetLastError() == hipSuccess ? 0 : 3;
}

extern "C" int inflight_copy_launch(const void *in, void *out, int64_t n_vec, int per_thread, void *stream) {
    const int64_t per_block = 256 * (int64_t)per_thread;
    if (n_vec % per_block) return 1;
    const dim3 grid((unsigned)(n_vec / per_block));
    switch (per_thread) {
        case 4: hipLaunchKernelGGL(inflight_copy<4>, grid, dim3(256), 0, (hipStream_t)stream, (const float4 *)in, (float4 *)out, n_vec); break;
        case 16: hipLaunchKernelGGL(inflight_copy<16>, grid, dim3(256), 0, (hipStream_t)stream, (const float4 *)in, (float4 *)out, n_vec); break;
        case 32: hipLaunchKernelGGL(inflight_copy<32>, grid, dim3(256), 0, (hipStream_t)stream, (const float4 *)in, (float4 *)out, n_vec); break;
        case 56: hipLaunchKernelGGL(inflight_copy<56>, grid, dim3(256), 0, (hipStream_t)stream, (const float4 *)in, (float4 *)out, n_vec); break;
        default: return 2;
    }
    return hipGetLastError() == hipSuccess ? 0 : 3;
}
