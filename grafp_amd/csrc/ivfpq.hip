// ivfpq.hip -- asymmetric-distance scan of an IVF-PQ index, gfx950.
//
// The published evaluation protocol searches with faiss IndexIVFPQ(IndexFlatL2(d), d, 64 lists, 64 sub-quantisers,
// 8 bits), nprobe = 20 (/root/reference/eval.py:65-69,122; the default of test_fp.py:276).  faiss==1.7.2 is not
// vendored; its published algorithm is restated: a vector is stored as the id of its nearest coarse centroid and, per
// sub-space of d/M dimensions, the id of the codeword nearest to the RESIDUAL (x - centroid); a query scans the codes
// of its nprobe nearest lists and estimates  ||q - x||^2 ~= sum_m ||(q - c)_m - codeword[m][code_m]||^2.
// Training (two k-means) and encoding are dense torch algebra in grafp_amd/ivfpq.py; this file is the scan:
//   one workgroup per (query, probed list): the (M x 256) table of sub-space distances of THIS residual goes to LDS
//   (64 KB at M = 64: two workgroups per CU), then the list's codes stream through once -- a thread owns a code, adds M
//   table entries (one ds_read_b32 each: the bank is set by the code byte, random), and writes (estimate, position);
//   the top-k selection over a query's concatenated lists and the id lookup stay with the caller.
// Bound: codes bytes from L2/HBM (M bytes per candidate) + M LDS reads per candidate.
#include "common.h"

namespace grafp {

template <int DSUB>
__global__ __launch_bounds__(256) void ivfpq_adc_kernel(const float *__restrict__ q, const float *__restrict__ centroids,
                                                        const float *__restrict__ codebooks,       // (M, 256, DSUB)
                                                        const unsigned char *__restrict__ codes,    // (n, M) in list order
                                                        const int64_t *__restrict__ list_start,     // (nlist + 1)
                                                        const int32_t *__restrict__ probe,          // (nq, nprobe)
                                                        const int64_t *__restrict__ out_start,      // (nq, nprobe)
                                                        int d, int M, int nprobe, int64_t row_stride,
                                                        float *__restrict__ out_dist, int32_t *__restrict__ out_pos) {
    extern __shared__ __attribute__((aligned(16))) float tab[];            // M x 256
    const int qi = blockIdx.y, slot = blockIdx.x, tid = threadIdx.x;
    const int list = probe[(size_t)qi * nprobe + slot];
    if (list < 0) return;
    const float *qv = q + (size_t)qi * d, *cv = centroids + (size_t)list * d;
    // table: thread = codeword id, loop over sub-spaces (codebook reads are coalesced over the codeword id)
    for (int m = 0; m < M; ++m) {
        float acc = 0.0f;
#pragma unroll
        for (int e = 0; e < DSUB; ++e) {
            const float r = qv[m * DSUB + e] - cv[m * DSUB + e];
            const float diff = r - codebooks[((size_t)m * 256 + tid) * DSUB + e];
            acc = __builtin_fmaf(diff, diff, acc);
        }
        tab[m * 256 + tid] = acc;
    }
    __syncthreads();
    const int64_t lo = list_start[list], len = list_start[list + 1] - lo;
    float *od = out_dist + (size_t)qi * row_stride + out_start[(size_t)qi * nprobe + slot];
    int32_t *op = out_pos + (size_t)qi * row_stride + out_start[(size_t)qi * nprobe + slot];
    for (int64_t i = tid; i < len; i += 256) {
        const unsigned char *c = codes + (size_t)(lo + i) * M;
        float acc = 0.0f;
        if ((M & 15) == 0) {
            for (int m0 = 0; m0 < M; m0 += 16) {
                const uint4 v = *reinterpret_cast<const uint4 *>(c + m0);
                const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int b = 0; b < 4; ++b) acc += tab[(m0 + 4 * u + b) * 256 + ((w[u] >> (8 * b)) & 255u)];
            }
        } else {
            for (int m = 0; m < M; ++m) acc += tab[m * 256 + c[m]];
        }
        od[i] = acc;
        op[i] = (int32_t)(lo + i);
    }
}

}  // namespace grafp

extern "C" int grafp_ivfpq_scan_f32(const float *q, int nq, int d, const float *centroids, int nlist,
                                    const float *codebooks, int M, const uint8_t *codes, const int64_t *list_start,
                                    const int32_t *probe, int nprobe, const int64_t *out_start, int64_t row_stride,
                                    float *out_dist, int32_t *out_pos, grafp_stream_t stream) {
    using namespace grafp;
    GRAFP_REQUIRE(q && centroids && codebooks && codes && list_start && probe && out_start && out_dist && out_pos,
                  "ivfpq_scan: null pointer");
    GRAFP_REQUIRE(nq > 0 && nq <= 65535 && nlist > 0 && nprobe > 0 && nprobe <= nlist && M > 0 && d % M == 0,
                  "ivfpq_scan: bad shape nq=%d nlist=%d nprobe=%d d=%d M=%d", nq, nlist, nprobe, d, M);
    const int dsub = d / M;
    GRAFP_REQUIRE(dsub == 1 || dsub == 2 || dsub == 4 || dsub == 8, "ivfpq_scan: d / M = %d not in {1, 2, 4, 8}", dsub);
    const size_t lds = (size_t)M * 256 * sizeof(float);
    GRAFP_REQUIRE(lds <= 160 * 1024, "ivfpq_scan: M = %d sub-quantisers need %zu bytes of LDS", M, lds);
    const dim3 grid(nprobe, nq);
    hipStream_t s = (hipStream_t)stream;
#define IVFPQ_LAUNCH(DS)                                                                                                 \
    do {                                                                                                                 \
        (void)hipFuncSetAttribute((const void *)ivfpq_adc_kernel<DS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((ivfpq_adc_kernel<DS>), grid, dim3(256), lds, s, q, centroids, codebooks, codes, list_start,     \
                           probe, out_start, d, M, nprobe, row_stride, out_dist, out_pos);                               \
    } while (0)
    if (dsub == 1) IVFPQ_LAUNCH(1);
    else if (dsub == 2) IVFPQ_LAUNCH(2);
    else if (dsub == 4) IVFPQ_LAUNCH(4);
    else IVFPQ_LAUNCH(8);
#undef IVFPQ_LAUNCH
    GRAFP_CHECK_LAUNCH("ivfpq_adc_kernel");
    return GRAFP_OK;
}
