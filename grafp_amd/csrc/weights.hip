// weights.hip -- all 1x1-convolution weights of the encoder prepared for one training step in ONE launch, gfx950.
//
// The reference keeps f32 Conv2d weights (/root/reference/encoder/gcn_lib/torch_nn.py:56-60, torch_vertex.py:152-162,
// encoder/graph_encoder.py:21-24,52-55) and lets cuDNN pick its own layouts.  The bf16 step here needs, per layer and
// per step, (a) the bf16 copy W (R, K/g) for the forward GEMM and (b) the per-group TRANSPOSED bf16 copy (g K/g, R/g)
// for the data-gradient GEMM -- for the first layer of a residual block with an identity block appended on the right,
// [W^T | I] (grafp_conv1x1_gemm_cat_bf16).  As torch ops that is a multi-tensor cast plus one transpose / concatenate
// kernel per layer (63 launches of ~5 us per step, which is 2-3 % of a 128-pair step); here one workgroup per 32 x 32
// tile of any layer reads the f32 tile once and writes both copies (the transposed one through LDS).
// The table is built once by the host (grafp_amd/ops.py: lowp_weights); the identity blocks are written once, too.
#include "common.h"
#include "dma_ring.h"

namespace grafp {

struct WeightEntry {            // eight 64-bit words per layer
    const float *src;           // (G * Rg, Kg) f32
    unsigned short *dst;        // (G * Rg, Kg) bf16
    unsigned short *dst_t;      // (G * Kg, ld_t) bf16: row g * Kg + k holds src[g * Rg + r][k] at column r
    int64_t Rg, Kg, G, ld_t, tile_base;
};

__global__ __launch_bounds__(256) void weights_prepare_kernel(const WeightEntry *__restrict__ table,
                                                              const int *__restrict__ tile_entry) {
    __shared__ unsigned short tile[32][34];
    const WeightEntry e = table[tile_entry[blockIdx.x]];
    const int t = (int)(blockIdx.x - e.tile_base);
    const int tiles_k = (int)(e.Kg / 32), per_group = (int)(e.Rg / 32) * tiles_k;
    const int g = t / per_group, rem = t - g * per_group;
    const int r0 = (rem / tiles_k) * 32, k0 = (rem % tiles_k) * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + ty + 8 * i;
        const size_t at = ((size_t)g * e.Rg + r) * e.Kg + k0 + tx;
        const unsigned short b = (unsigned short)(gm_pack_bf16(e.src[at], 0.0f) & 0xffffu);     // round to nearest even
        e.dst[at] = b;
        tile[ty + 8 * i][tx] = b;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = ty + 8 * i;
        e.dst_t[((size_t)g * e.Kg + k0 + k) * e.ld_t + r0 + tx] = tile[tx][k];
    }
}

}  // namespace grafp

extern "C" int grafp_weights_prepare(const void *table, const int32_t *tile_entry, int n_tiles, grafp_stream_t stream) {
    using namespace grafp;
    GRAFP_REQUIRE(table && tile_entry && n_tiles > 0, "weights_prepare: bad arguments");
    hipLaunchKernelGGL(weights_prepare_kernel, dim3(n_tiles), dim3(256), 0, (hipStream_t)stream,
                       (const WeightEntry *)table, tile_entry);
    GRAFP_CHECK_LAUNCH("weights_prepare_kernel");
    return GRAFP_OK;
}
