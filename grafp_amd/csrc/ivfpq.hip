// ivfpq.hip -- asymmetric-distance scan of an IVF-PQ index, gfx950.
//
// The published evaluation protocol searches with faiss IndexIVFPQ(IndexFlatL2(d), d, 64 lists, 64 sub-quantisers,
// 8 bits), nprobe = 20 (/root/reference/eval.py:65-69,122; the default of test_fp.py:276).  faiss==1.7.2 is not
// vendored; its published algorithm is restated: a vector is stored as the id of its nearest coarse centroid and, per
// sub-space of d/M dimensions, the id of the codeword nearest to the RESIDUAL (x - centroid); a query scans the codes
// of its nprobe nearest lists and estimates  ||q - x||^2 ~= sum_m ||(q - c)_m - codeword[m][code_m]||^2.
// Everything the index does runs here (round 4; rounds 1-3 had the scan only and torch algebra around it):
//   pq_assign_kernel      nearest centroid / codeword of every (row, sub-space) -- coarse assignment (one sub-space of d
//                         dims, nlist centroids) and PQ encoding (M sub-spaces of d/M dims, 256 codewords, residuals
//                         taken on the fly) alike; squared distances as fmaf chains over the dims, lowest id on ties;
//   kmeans_*_kernel       seeded Lloyd iterations on the device: assignment (above), per-chunk cluster sums in row order,
//                         chunk sums in chunk order (a FIXED summation order: reproducible, restated bit for bit by
//                         oracle/csrc/ivfpq.c), centroid = sum / count, empty clusters keep their centroid;
//   ivfpq_probe_kernel    the nprobe nearest lists of a query by (distance, list id);
//   ivfpq_search_kernel   one workgroup per query: per probed list the (M x 256) table of sub-space distances of THIS
//                         residual goes to LDS (64 KB at M = 64: two workgroups per CU), the list's codes stream through
//                         once -- a thread owns a code and adds M table entries (one ds_read_b32 each: the bank is set by
//                         the code byte, random) -- and the estimates go straight into the wave's running top-k (topk.h):
//                         no (query x probed codes) scratch, no host round trip, ids out;
//   ivfpq_adc_kernel      the round-1 form of the scan (dense estimates out; grafp_ivfpq_scan_f32): a caller that wants
//                         more than GRAFP_SEARCH_MAX_K = 32 results per query selects from it itself (IVFPQIndex.search
//                         serves k <= 32, what eval.py:122,269 asks for: k_probe = 20).
// Bound of a search: codes bytes from L2/HBM (M bytes per candidate) + M LDS reads per candidate.
#include "common.h"
#include "topk.h"

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

// ---- nearest centroid per (row, sub-space) -------------------------------------------------------------------------
// x (n, D) f32, G sub-spaces of d = D / G dims; optional residual: r = x[row] - base[base_idx[row]] (one rounding), then
// dist_j = fmaf chain over c of (r_c - cent[g][j][c])^2 -- the same two roundings per term as the search's table.
// Grid (row blocks, G); the sub-space's centroids pass through LDS (all k x d of them when they fit 64 KB, otherwise kb at
// a time: IndexIVFPQ with 512 or 1024 lists at d = 128), every thread owns a row and reads them as broadcasts.
template <int DCAP>
__global__ __launch_bounds__(256) void pq_assign_kernel(const float *__restrict__ x, int64_t n, int D, int G,
                                                        const float *__restrict__ base,
                                                        const int32_t *__restrict__ base_idx,
                                                        const float *__restrict__ cent, int k, int kb,
                                                        int32_t *__restrict__ out, uint8_t *__restrict__ out_u8) {
    extern __shared__ __attribute__((aligned(16))) float sc[];             // kb x d: one tile of the centroids at a time
    const int d = D / G, g = blockIdx.y, tid = threadIdx.x;
    const float *cg = cent + (size_t)g * k * d;
    const int64_t row = (int64_t)blockIdx.x * 256 + tid;
    const bool live = row < n;
    float r[DCAP];
    if (live) {
        const float *xr = x + row * D + (size_t)g * d;
        const float *br = base ? base + (size_t)base_idx[row] * D + (size_t)g * d : nullptr;
#pragma unroll
        for (int c = 0; c < DCAP; ++c)
            if (c < d) r[c] = br ? xr[c] - br[c] : xr[c];
    }
    float best = INFINITY;
    int bj = 0;
    // any k: the centroids pass through LDS kb at a time, in id order -- the running (best, id) sees them in the same
    // order as one resident table would (strict <: lowest id on ties), so the result does not depend on kb
    for (int j0 = 0; j0 < k; j0 += kb) {
        const int kt = (k - j0) < kb ? (k - j0) : kb;
        __syncthreads();                                                   // the previous tile has been read by everybody
        for (int i = tid; i < kt * d; i += 256) sc[i] = cg[(size_t)j0 * d + i];
        __syncthreads();
        if (live) {
            for (int j = 0; j < kt; ++j) {
                const float *cj = sc + j * d;
                float acc = 0.0f;
#pragma unroll
                for (int c = 0; c < DCAP; ++c)
                    if (c < d) {
                        const float diff = r[c] - cj[c];
                        acc = __builtin_fmaf(diff, diff, acc);
                    }
                if (acc < best) {                  // strict: lowest id on ties
                    best = acc;
                    bj = j0 + j;
                }
            }
        }
    }
    if (!live) return;
    if (out) out[row * G + g] = bj;
    if (out_u8) out_u8[row * G + g] = (uint8_t)bj;
}

// ---- k-means pieces ---------------------------------------------------------------------------------------------------
constexpr int KM_CHUNK = 1024;           // rows per partial sum

// centroid j of sub-space g <- residual of training row init_rows[j]
__global__ __launch_bounds__(256) void kmeans_init_kernel(const float *__restrict__ x, int D, int G,
                                                          const float *__restrict__ base,
                                                          const int32_t *__restrict__ base_idx,
                                                          const int64_t *__restrict__ init_rows, int k,
                                                          float *__restrict__ cent) {
    const int d = D / G;
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;            // over (g, j, c)
    if (e >= (int64_t)G * k * d) return;
    const int c = (int)(e % d), j = (int)((e / d) % k), g = (int)(e / ((int64_t)d * k));
    const int64_t row = init_rows[j];
    const float v = x[row * D + g * d + c];
    cent[e] = base ? v - base[(size_t)base_idx[row] * D + g * d + c] : v;
}

// partial[chunk][g][j][c] = sum over the chunk's rows assigned to j, in row order; pcnt[chunk][g][j] = their number.
// Grid (chunks, G).  Thread t owns the outputs e = t, t + 256, ... of the k x d block and walks the rows itself: no
// atomics, so the sum has ONE order.
__global__ __launch_bounds__(256) void kmeans_partial_kernel(const float *__restrict__ x, int64_t n, int D, int G,
                                                             const float *__restrict__ base,
                                                             const int32_t *__restrict__ base_idx,
                                                             const int32_t *__restrict__ asg, int k,
                                                             float *__restrict__ partial, int32_t *__restrict__ pcnt) {
    __shared__ int s_asg[KM_CHUNK];
    __shared__ int s_base[KM_CHUNK];
    const int d = D / G, g = blockIdx.y, tid = threadIdx.x;
    const int64_t lo = (int64_t)blockIdx.x * KM_CHUNK;
    const int rows = (int)((n - lo) < KM_CHUNK ? (n - lo) : KM_CHUNK);
    for (int i = tid; i < rows; i += 256) {
        s_asg[i] = asg[(lo + i) * G + g];
        s_base[i] = base ? base_idx[lo + i] : 0;
    }
    __syncthreads();
    float *po = partial + ((size_t)blockIdx.x * G + g) * k * d;
    int32_t *pc = pcnt + ((size_t)blockIdx.x * G + g) * k;
    for (int e = tid; e < k * d; e += 256) {
        const int j = e / d, c = e - j * d;
        float s = 0.0f;
        int cnt = 0;
        for (int i = 0; i < rows; ++i) {
            if (s_asg[i] == j) {
                const float v = x[(lo + i) * D + g * d + c];
                s += base ? v - base[(size_t)s_base[i] * D + g * d + c] : v;
                ++cnt;
            }
        }
        po[e] = s;
        if (c == 0) pc[j] = cnt;
    }
}

// cent[g][j][c] <- (sum over chunks, in chunk order) / count; a cluster without rows keeps its centroid
__global__ __launch_bounds__(256) void kmeans_update_kernel(const float *__restrict__ partial,
                                                            const int32_t *__restrict__ pcnt, int nchunks, int G, int k,
                                                            int d, float *__restrict__ cent) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;            // over (g, j, c)
    if (e >= (int64_t)G * k * d) return;
    const int64_t gj = e / d;
    float s = 0.0f;
    int cnt = 0;
    for (int ch = 0; ch < nchunks; ++ch) {
        s += partial[(size_t)ch * G * k * d + e];
        cnt += pcnt[(size_t)ch * G * k + gj];
    }
    if (cnt > 0) cent[e] = s / (float)cnt;
}

// ---- the nprobe nearest lists of a query: one wave per query ------------------------------------------------------------
__global__ __launch_bounds__(64) void ivfpq_probe_kernel(const float *__restrict__ q, int nq, int d,
                                                         const float *__restrict__ cent, int nlist, int nprobe,
                                                         int32_t *__restrict__ probe) {
    extern __shared__ __attribute__((aligned(16))) float sdist[];         // nlist
    const int qi = blockIdx.x, lane = threadIdx.x;
    const float *qv = q + (size_t)qi * d;
    for (int l = lane; l < nlist; l += 64) {
        const float *cv = cent + (size_t)l * d;
        float acc = 0.0f;
        for (int c = 0; c < d; ++c) {
            const float diff = qv[c] - cv[c];
            acc = __builtin_fmaf(diff, diff, acc);
        }
        sdist[l] = acc;
    }
    WAVE_SYNC();
    for (int s = 0; s < nprobe; ++s) {
        float bd = INFINITY;
        int bl = SR_EMPTY;
        for (int l = lane; l < nlist; l += 64) {
            const float v = sdist[l];
            if (lex_lt(v, l, bd, bl)) {
                bd = v;
                bl = l;
            }
        }
#pragma unroll
        for (int j = 1; j < 64; j <<= 1) {
            const float od = __shfl_xor(bd, j);
            const int ol = __shfl_xor(bl, j);
            if (lex_lt(od, ol, bd, bl)) {
                bd = od;
                bl = ol;
            }
        }
        if (lane == 0) {
            probe[(size_t)qi * nprobe + s] = bl == SR_EMPTY ? -1 : bl;
            if (bl != SR_EMPTY) sdist[bl] = __builtin_nanf("");          // taken: NaN loses every comparison
        }
        WAVE_SYNC();
    }
}

// ---- search: scan of the probed lists with the running top-k fused in ---------------------------------------------------
template <int DSUB>
__global__ __launch_bounds__(256) void ivfpq_search_kernel(const float *__restrict__ q, const float *__restrict__ centroids,
                                                           const float *__restrict__ codebooks,      // (M, 256, DSUB)
                                                           const unsigned char *__restrict__ codes,   // (n, M), list order
                                                           const int64_t *__restrict__ list_start,    // (nlist + 1)
                                                           const int64_t *__restrict__ ids,           // (n) position -> id
                                                           const int32_t *__restrict__ probe, int d, int M, int nprobe,
                                                           int k, float *__restrict__ out_d, int64_t *__restrict__ out_i) {
    extern __shared__ __attribute__((aligned(16))) float tab[];           // M x 256
    __shared__ float pend_d[4][WT_PEND];
    __shared__ int pend_i[4][WT_PEND];
    __shared__ float wtop_d[4][32];
    __shared__ int wtop_i[4][32];
    const int qi = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const float *qv = q + (size_t)qi * d;
    WaveTop top;
    top.init(pend_d[wave], pend_i[wave], INFINITY);
    for (int slot = 0; slot < nprobe; ++slot) {
        const int list = probe[(size_t)qi * nprobe + slot];
        if (list < 0) continue;                                            // uniform
        const int64_t lo = list_start[list], len = list_start[list + 1] - lo;
        if (len == 0) continue;
        const float *cv = centroids + (size_t)list * d;
        __syncthreads();                                                   // the previous list's table is done with
        for (int m = 0; m < M; ++m) {                                     // thread = codeword id
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
        for (int64_t i0 = 0; i0 < len; i0 += 256) {
            const int64_t i = i0 + tid;
            const bool valid = i < len;
            float acc = INFINITY;
            int id = SR_EMPTY;
            if (valid) {
                const unsigned char *c = codes + (size_t)(lo + i) * M;
                acc = 0.0f;
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
                id = (int)ids[lo + i];
            }
            top.push(valid, acc, id, k, lane);                             // wave-level call (i0 is uniform)
        }
    }
    if (top.pc > 0) top.fold(k, lane);
    float td;
    int ti;
    block_merge_tops(top, wtop_d, wtop_i, wave, lane, td, ti);
    if (wave == 0 && lane < k) {
        out_d[(size_t)qi * k + lane] = td;
        out_i[(size_t)qi * k + lane] = ti == SR_EMPTY ? (int64_t)-1 : (int64_t)ti;
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

extern "C" int grafp_pq_assign_f32(const float *x, int64_t n, int D, int G, const float *base, const int32_t *base_idx,
                                   const float *cent, int k, int32_t *out, uint8_t *out_u8, grafp_stream_t stream) {
    using namespace grafp;
    GRAFP_REQUIRE(x && cent && (out || out_u8), "pq_assign: null pointer");
    GRAFP_REQUIRE((base == nullptr) == (base_idx == nullptr), "pq_assign: base and base_idx go together");
    GRAFP_REQUIRE(n >= 0 && D > 0 && G > 0 && D % G == 0 && k >= 1, "pq_assign: bad shape n=%lld D=%d G=%d k=%d",
                  (long long)n, D, G, k);
    GRAFP_REQUIRE(!out_u8 || k <= 256, "pq_assign: byte codes need k <= 256 (k=%d)", k);
    GRAFP_REQUIRE(G <= 65535, "pq_assign: G=%d sub-spaces", G);
    const int d = D / G;
    GRAFP_REQUIRE(d <= 128, "pq_assign: %d dims per sub-space (<= 128)", d);
    // centroids per LDS tile: all of them up to 64 KB (two workgroups per CU), otherwise 64 KB worth
    int kb = (64 * 1024) / (int)(d * sizeof(float));
    if (kb > k) kb = k;
    const size_t lds = (size_t)kb * d * sizeof(float);
    if (n == 0) return GRAFP_OK;
    const dim3 grid((unsigned)((n + 255) / 256), G);
    hipStream_t s = (hipStream_t)stream;
#define PQA_LAUNCH(DC)                                                                                                   \
    do {                                                                                                                 \
        (void)hipFuncSetAttribute((const void *)pq_assign_kernel<DC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((pq_assign_kernel<DC>), grid, dim3(256), lds, s, x, n, D, G, base, base_idx, cent, k, kb, out, out_u8); \
    } while (0)
    if (d <= 2) PQA_LAUNCH(2);
    else if (d <= 8) PQA_LAUNCH(8);
    else if (d <= 32) PQA_LAUNCH(32);
    else PQA_LAUNCH(128);
#undef PQA_LAUNCH
    GRAFP_CHECK_LAUNCH("pq_assign_kernel");
    return GRAFP_OK;
}

static size_t km_align(size_t v) { return (v + 255) & ~(size_t)255; }

extern "C" size_t grafp_kmeans_workspace(int64_t n, int D, int G, int k) {
    if (n <= 0 || D <= 0 || G <= 0 || D % G || k <= 0) return 0;
    const size_t nchunks = (size_t)((n + grafp::KM_CHUNK - 1) / grafp::KM_CHUNK);
    return km_align((size_t)n * G * sizeof(int32_t)) + km_align(nchunks * G * k * (D / G) * sizeof(float)) +
           km_align(nchunks * G * k * sizeof(int32_t));
}

extern "C" int grafp_kmeans_f32(const float *x, int64_t n, int D, int G, const float *base, const int32_t *base_idx,
                                const int64_t *init_rows, int k, int niter, float *cent, void *ws, size_t ws_bytes,
                                grafp_stream_t stream) {
    using namespace grafp;
    GRAFP_REQUIRE(x && init_rows && cent, "kmeans: null pointer");
    GRAFP_REQUIRE((base == nullptr) == (base_idx == nullptr), "kmeans: base and base_idx go together");
    GRAFP_REQUIRE(n >= 1 && D > 0 && G > 0 && D % G == 0 && k >= 1 && niter >= 0 && G <= 65535,
                  "kmeans: bad shape n=%lld D=%d G=%d k=%d niter=%d", (long long)n, D, G, k, niter);
    const size_t need = grafp_kmeans_workspace(n, D, G, k);
    if (!ws || ws_bytes < need) {
        set_error("kmeans: workspace %zu bytes < required %zu", ws_bytes, need);
        return GRAFP_ERR_WORKSPACE;
    }
    const int d = D / G;
    const int nchunks = (int)((n + KM_CHUNK - 1) / KM_CHUNK);
    char *w = (char *)ws;
    int32_t *asg = (int32_t *)w;            w += km_align((size_t)n * G * sizeof(int32_t));
    float *partial = (float *)w;            w += km_align((size_t)nchunks * G * k * d * sizeof(float));
    int32_t *pcnt = (int32_t *)w;
    hipStream_t s = (hipStream_t)stream;
    const int64_t ne = (int64_t)G * k * d;
    hipLaunchKernelGGL(kmeans_init_kernel, dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, s, x, D, G, base, base_idx,
                       init_rows, k, cent);
    GRAFP_CHECK_LAUNCH("kmeans_init_kernel");
    for (int it = 0; it < niter; ++it) {
        const int rc = grafp_pq_assign_f32(x, n, D, G, base, base_idx, cent, k, asg, nullptr, stream);
        if (rc != GRAFP_OK) return rc;
        hipLaunchKernelGGL(kmeans_partial_kernel, dim3(nchunks, G), dim3(256), 0, s, x, n, D, G, base, base_idx,
                           (const int32_t *)asg, k, partial, pcnt);
        hipLaunchKernelGGL(kmeans_update_kernel, dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, s,
                           (const float *)partial, (const int32_t *)pcnt, nchunks, G, k, d, cent);
        GRAFP_CHECK_LAUNCH("kmeans_partial_kernel / kmeans_update_kernel");
    }
    return GRAFP_OK;
}

extern "C" int grafp_ivfpq_probe_f32(const float *q, int nq, int d, const float *centroids, int nlist, int nprobe,
                                     int32_t *probe, grafp_stream_t stream) {
    using namespace grafp;
    GRAFP_REQUIRE(q && centroids && probe, "ivfpq_probe: null pointer");
    GRAFP_REQUIRE(nq >= 0 && d > 0 && nlist > 0 && nlist <= 16384 && nprobe > 0 && nprobe <= nlist,
                  "ivfpq_probe: bad shape nq=%d d=%d nlist=%d nprobe=%d", nq, d, nlist, nprobe);
    if (nq == 0) return GRAFP_OK;
    hipLaunchKernelGGL(ivfpq_probe_kernel, dim3(nq), dim3(64), (size_t)nlist * sizeof(float), (hipStream_t)stream, q, nq, d,
                       centroids, nlist, nprobe, probe);
    GRAFP_CHECK_LAUNCH("ivfpq_probe_kernel");
    return GRAFP_OK;
}

extern "C" int grafp_ivfpq_search_f32(const float *q, int nq, int d, const float *centroids, int nlist,
                                      const float *codebooks, int M, const uint8_t *codes, const int64_t *list_start,
                                      const int64_t *ids, const int32_t *probe, int nprobe, int k, float *out_dist,
                                      int64_t *out_ids, grafp_stream_t stream) {
    using namespace grafp;
    GRAFP_REQUIRE(q && centroids && codebooks && codes && list_start && ids && probe && out_dist && out_ids,
                  "ivfpq_search: null pointer");
    GRAFP_REQUIRE(nq >= 0 && nlist > 0 && nprobe > 0 && nprobe <= nlist && M > 0 && d % M == 0,
                  "ivfpq_search: bad shape nq=%d nlist=%d nprobe=%d d=%d M=%d", nq, nlist, nprobe, d, M);
    GRAFP_REQUIRE(k >= 1 && k <= GRAFP_SEARCH_MAX_K, "ivfpq_search: k=%d not in [1, %d]", k, GRAFP_SEARCH_MAX_K);
    const int dsub = d / M;
    GRAFP_REQUIRE(dsub == 1 || dsub == 2 || dsub == 4 || dsub == 8, "ivfpq_search: d / M = %d not in {1, 2, 4, 8}", dsub);
    const size_t lds = (size_t)M * 256 * sizeof(float);
    GRAFP_REQUIRE(lds <= 150 * 1024, "ivfpq_search: M = %d sub-quantisers need %zu bytes of LDS", M, lds);
    if (nq == 0) return GRAFP_OK;
    hipStream_t s = (hipStream_t)stream;
#define IVFPQ_SEARCH(DS)                                                                                                 \
    do {                                                                                                                 \
        (void)hipFuncSetAttribute((const void *)ivfpq_search_kernel<DS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((ivfpq_search_kernel<DS>), dim3(nq), dim3(256), lds, s, q, centroids, codebooks, codes,        \
                           list_start, ids, probe, d, M, nprobe, k, out_dist, out_ids);                                   \
    } while (0)
    if (dsub == 1) IVFPQ_SEARCH(1);
    else if (dsub == 2) IVFPQ_SEARCH(2);
    else if (dsub == 4) IVFPQ_SEARCH(4);
    else IVFPQ_SEARCH(8);
#undef IVFPQ_SEARCH
    GRAFP_CHECK_LAUNCH("ivfpq_search_kernel");
    return GRAFP_OK;
}
