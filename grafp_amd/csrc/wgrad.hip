// wgrad.hip -- weight gradient of a 1x1 convolution on the (C, M = B*N) layout, gfx950.
//
//   dW[o][c] = sum_m G[o][m] * X[c][m]        G = dL/dY (Cout x M), X = layer input (Cin x M), both bf16, rows
//                                             contiguous; M = B*N = 262 144 ... 32 768 at B = 256
// This is the backward of every 1x1 Conv2d of the encoder (/root/reference/encoder/gcn_lib/torch_vertex.py:152-162,
// torch_nn.py:56, encoder/graph_encoder.py:52-55,131,156) w.r.t. its weight.  The output is tiny (64x64 ... 2048x512)
// and the contraction is enormous, with BOTH operands contiguous along the contraction: the library GEMM reaches
// ~10 TFLOP/s here (80-210 us where the operands stream in 13-34 us).  It is a streaming reduction:
//   * split-K: blockIdx.x owns a slice of M; blockIdx.y a 64x64 output tile; blockIdx.z a conv group;
//   * 128-column chunks of the G and X tiles are staged in LDS with coalesced 16-byte loads;
//   * each of the 4 waves owns a 32x32 quadrant: A and B fragments of v_mfma_f32_32x32x16_bf16 are 16 contiguous
//     bytes of a G / X row (8 consecutive m), read with ds_read_b128 from rows padded to 272 B (conflict-free);
//   * partial tiles go to a (S, Cout, Cin/g) f32 scratch, summed by wgrad_reduce_kernel (deterministic, no atomics).
// HBM-bound: (Cout + Cin) * M * 2 bytes per launch (+ re-reads of the smaller operand across output tiles).
#include "common.h"

namespace grafp {

typedef short bf16x8 __attribute__((ext_vector_type(8)));

constexpr int WG_T = 64;                 // output tile edge
constexpr int WG_KC = 128;               // contraction columns per LDS chunk
constexpr int WG_LS = WG_KC * 2 + 16;    // LDS row stride in bytes (272: ds_read_b128 rows land on distinct bank quads)

__global__ __launch_bounds__(256) void wgrad_partial_kernel(const unsigned short *__restrict__ G,
                                                            const unsigned short *__restrict__ X, int64_t M,
                                                            int cout_g, int cin_g, int tiles_c, int64_t cols_per_split,
                                                            float *__restrict__ part) {
    __shared__ __attribute__((aligned(16))) unsigned char sG[WG_T * WG_LS];
    __shared__ __attribute__((aligned(16))) unsigned char sX[WG_T * WG_LS];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    const int split = blockIdx.x, tile = blockIdx.y, grp = blockIdx.z;
    const int o0 = (tile / tiles_c) * WG_T, c0 = (tile % tiles_c) * WG_T;
    const unsigned short *Gg = G + (size_t)grp * cout_g * M;
    const unsigned short *Xg = X + (size_t)grp * cin_g * M;
    const int64_t m_begin = (int64_t)split * cols_per_split;
    const int64_t m_end = (m_begin + cols_per_split < M) ? m_begin + cols_per_split : M;
    const int wo = wave >> 1, wc = wave & 1;

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;

    const bool vec_ok = (M & 7) == 0;
    for (int64_t m0 = m_begin; m0 < m_end; m0 += WG_KC) {
        __syncthreads();
        // stage 64 rows x 128 columns of G and X: thread -> (row = i*16 + tid/16, 16-byte column tid%16)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = i * 16 + (tid >> 4), cb = tid & 15;
            const int64_t m = m0 + cb * 8;
            uint4 vg = make_uint4(0, 0, 0, 0), vx = make_uint4(0, 0, 0, 0);
            if (vec_ok && m + 8 <= m_end) {
                if (o0 + row < cout_g) vg = *reinterpret_cast<const uint4 *>(Gg + (size_t)(o0 + row) * M + m);
                if (c0 + row < cin_g) vx = *reinterpret_cast<const uint4 *>(Xg + (size_t)(c0 + row) * M + m);
            } else if (m < m_end) {
                unsigned short tg[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tx[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                for (int e = 0; e < 8 && m + e < m_end; ++e) {
                    if (o0 + row < cout_g) tg[e] = Gg[(size_t)(o0 + row) * M + m + e];
                    if (c0 + row < cin_g) tx[e] = Xg[(size_t)(c0 + row) * M + m + e];
                }
                vg = make_uint4(tg[0] | (tg[1] << 16), tg[2] | (tg[3] << 16), tg[4] | (tg[5] << 16), tg[6] | (tg[7] << 16));
                vx = make_uint4(tx[0] | (tx[1] << 16), tx[2] | (tx[3] << 16), tx[4] | (tx[5] << 16), tx[6] | (tx[7] << 16));
            }
            *reinterpret_cast<uint4 *>(sG + row * WG_LS + cb * 16) = vg;
            *reinterpret_cast<uint4 *>(sX + row * WG_LS + cb * 16) = vx;
        }
        __syncthreads();
        const unsigned char *ga = sG + (wo * 32 + l31) * WG_LS + half * 16;
        const unsigned char *xa = sX + (wc * 32 + l31) * WG_LS + half * 16;
#pragma unroll
        for (int kk = 0; kk < WG_KC / 16; ++kk) {
            const bf16x8 a = *reinterpret_cast<const bf16x8 *>(ga + kk * 32);
            const bf16x8 b = *reinterpret_cast<const bf16x8 *>(xa + kk * 32);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
        }
    }
    // partial tile -> part[split][grp][o][c]
    float *pp = part + ((size_t)split * gridDim.z + grp) * cout_g * cin_g;
    const int c = c0 + wc * 32 + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int o = o0 + wo * 32 + mfma_row(r, half);
        if (o < cout_g && c < cin_g) pp[(size_t)o * cin_g + c] = acc[r];
    }
}

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float *__restrict__ part, int S, int64_t n,
                                                           float *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float s = 0.0f;
    for (int k = 0; k < S; ++k) s += part[(size_t)k * n + i];
    out[i] = s;
}

struct WgradPlan {
    int tiles_o, tiles_c, S;
    int64_t cols;
};
static WgradPlan wgrad_plan(int cout_g, int cin_g, int groups, int64_t M) {
    WgradPlan p;
    p.tiles_o = (cout_g + WG_T - 1) / WG_T;
    p.tiles_c = (cin_g + WG_T - 1) / WG_T;
    const int64_t tiles = (int64_t)p.tiles_o * p.tiles_c * groups;
    int64_t S = (2048 + tiles - 1) / tiles;                 // ~2048 workgroups: 8 per CU keeps loads in flight
    const int64_t max_s = (M + 4 * WG_KC - 1) / (4 * WG_KC);  // >= 4 chunks per slice
    if (S > max_s) S = max_s;
    if (S < 1) S = 1;
    int64_t cols = (M + S - 1) / S;
    cols = (cols + WG_KC - 1) / WG_KC * WG_KC;
    p.cols = cols;
    p.S = (int)((M + cols - 1) / cols);
    return p;
}

}  // namespace grafp

extern "C" size_t grafp_conv1x1_wgrad_workspace(int Cout, int Cin, int groups, int64_t M) {
    using namespace grafp;
    if (Cout <= 0 || Cin <= 0 || groups <= 0 || M <= 0 || Cout % groups || Cin % groups) return 0;
    const WgradPlan p = wgrad_plan(Cout / groups, Cin / groups, groups, M);
    return (size_t)p.S * Cout * (Cin / groups) * sizeof(float);
}

extern "C" int grafp_conv1x1_wgrad_bf16(const void *grad_out, const void *x, int Cout, int Cin, int groups, int64_t M,
                                        float *dweight, void *ws, size_t ws_bytes, grafp_stream_t stream) {
    using namespace grafp;
    GRAFP_REQUIRE(grad_out && x && dweight, "conv1x1_wgrad: null pointer");
    GRAFP_REQUIRE(Cout > 0 && Cin > 0 && groups > 0 && M > 0 && Cout % groups == 0 && Cin % groups == 0,
                  "conv1x1_wgrad: bad shape Cout=%d Cin=%d groups=%d M=%lld", Cout, Cin, groups, (long long)M);
    GRAFP_REQUIRE((((uintptr_t)grad_out | (uintptr_t)x) & 15) == 0, "conv1x1_wgrad: operands must be 16-byte aligned");
    const size_t need = grafp_conv1x1_wgrad_workspace(Cout, Cin, groups, M);
    if (!ws || ws_bytes < need) {
        set_error("conv1x1_wgrad: workspace %zu bytes < required %zu", ws_bytes, need);
        return GRAFP_ERR_WORKSPACE;
    }
    const int cout_g = Cout / groups, cin_g = Cin / groups;
    const WgradPlan p = wgrad_plan(cout_g, cin_g, groups, M);
    GRAFP_REQUIRE(p.tiles_o * p.tiles_c <= 65535, "conv1x1_wgrad: output too large");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(wgrad_partial_kernel, dim3(p.S, p.tiles_o * p.tiles_c, groups), dim3(256), 0, s,
                       (const unsigned short *)grad_out, (const unsigned short *)x, M, cout_g, cin_g, p.tiles_c, p.cols,
                       (float *)ws);
    GRAFP_CHECK_LAUNCH("wgrad_partial_kernel");
    const int64_t n = (int64_t)Cout * cin_g;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const float *)ws, p.S, n,
                       dweight);
    GRAFP_CHECK_LAUNCH("wgrad_reduce_kernel");
    return GRAFP_OK;
}
