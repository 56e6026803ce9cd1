/* grafp_hip.h -- C ABI of libgrafp_hip.so, the MI355X (gfx950) hot path of GraFPrint.
 *
 * The reference (chymaera96/GraFP) is pure Python/PyTorch and has no FFI: its "operator interface"
 * for this path is a set of torch calls inside Python modules.  Each entry point below replaces one
 * such call site (cited per function, paths relative to the reference root) and is bound from
 * Python with ctypes (grafp_amd/_lib.py; INTEGRATION.md shows the reference-side stubs).
 *
 * Conventions (all entry points):
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless it says "host";
 *   - the CALLER owns every buffer, including the scratch `ws` whose size the matching
 *     *_workspace() function returns (bytes; 256-byte alignment is sufficient);
 *   - no allocation, no global mutable state, no environment variable read (the launch plan of a call is a pure
 *     function of its arguments), no implicit synchronisation: work is enqueued on
 *     `stream` (a hipStream_t passed as void*; NULL = the default stream) and is hipGraph-capturable;
 *   - returns GRAFP_OK (0) or a negative GRAFP_ERR_* code; grafp_last_error() then holds a
 *     thread-local human-readable message;
 *   - layouts are the reference's: activations (B, C, N) channel-major f32 (the reference's
 *     (B,C,N,1) with the trailing 1 dropped), edge indices int64, fingerprints (n, d) row-major f32.
 */
#ifndef GRAFP_HIP_H
#define GRAFP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GRAFP_OK 0
#define GRAFP_ERR_ARG (-1)       /* bad argument (shape, alignment, unsupported size) */
#define GRAFP_ERR_WORKSPACE (-2) /* workspace missing or too small */
#define GRAFP_ERR_LAUNCH (-3)    /* HIP launch failed (message holds hipGetErrorString) */

#define GRAFP_ABI_VERSION 1
#define GRAFP_KNN_MAX_K 8     /* edges per node supported by grafp_knn_graph_f32 */
#define GRAFP_SEARCH_MAX_K 32 /* results per query supported by grafp_knn_search_l2_f32 */

typedef void *grafp_stream_t; /* hipStream_t */

/* element types of the *_strided entry points (bf16 = upper 16 bits of an IEEE f32, round-to-nearest-even) */
#define GRAFP_F32 0
#define GRAFP_BF16 1

int grafp_abi_version(void);
const char *grafp_last_error(void);

/* ---- K1: log-mel spectrogram ---------------------------------------------------------------
 * Replaces torchaudio MelSpectrogram(sample_rate, win_length, hop_length, n_fft, n_mels) ->
 * AmplitudeToDB() as constructed at modules/transformations.py:50-57 and applied at :78,:83
 * (batched 1 s clips) and :89,:111 (whole track).  Centred reflect-padded STFT (radix-2 FFT in LDS,
 * two real frames per complex transform), |.|^2, sparse triangular mel filterbank, 10*log10(max(.,1e-10)).
 *   wav      (B, T) f32, row stride wav_stride elements          T > n_fft/2
 *   window   (n_fft) f32       analysis window, already zero-padded to n_fft
 *   twiddle  (n_fft/2, 2) f32  (cos, -sin)(2*pi*j/n_fft)
 *   fb       (n_fft/2+1, n_mels) f32 dense filterbank; band_lo/band_hi (n_mels) int32 = first/last
 *            bin with a non-zero weight per band (band_hi < band_lo for an empty band)
 *   out      (B, n_mels, n_frames) f32, n_frames = 1 + T / hop
 * n_fft in {256, 512, 1024, 2048}. */
int grafp_logmel_f32(const float *wav, int64_t wav_stride, int B, int T, int n_fft, int hop, int n_mels,
                     const float *window, const float *twiddle, const float *fb, const int32_t *band_lo,
                     const int32_t *band_hi, float *out, grafp_stream_t stream);

/* Whole-track segmentation, modules/transformations.py:89-90 (`transpose(1,0).unfold(0, size, step)`),
 * materialised contiguously: spec (n_mels, n_frames) -> seg (n_seg, n_mels, size),
 * n_seg = (n_frames - size) / step + 1. */
int grafp_unfold_segments_f32(const float *spec, int n_mels, int n_frames, int size, int step, float *seg,
                              grafp_stream_t stream);

/* ---- K2: peak extractor ----------------------------------------------------------------------
 * Replaces GPUPeakExtractorv2.forward, peak_extractor.py:56-82: per-clip min-max normalisation,
 * [T-ramp, F-ramp, spec] stack, Conv2d(3 -> F, (KH,KW), stride (sh,1), pad (KH/2,KW/2)) + ReLU, flatten.
 *   spec (B,H,W)  weight (F,3,KH,KW)  bias (F)  out (B,F,Ho*W), Ho = (H + 2*(KH/2) - KH)/sh + 1
 *   t_ramp (W) = linspace(0,1,W), f_ramp (H) = linspace(0,1,H)   (peak_extractor.py:36-42), KH, KW odd
 * Backward (spec carries no gradient, train.py:66-67): dweight (F,3,KH,KW) / dbias (F) are WRITTEN (not accumulated
 * into); `out` is the forward result (ReLU mask).  Bit-reproducible: per-workgroup partial sums in the workspace
 * (grafp_peak_extract_bwd_workspace bytes), added in workgroup order by a second launch -- no float atomics. */
int grafp_peak_extract_fwd_f32(const float *spec, int B, int H, int W, const float *weight, const float *bias,
                               int F, int KH, int KW, int stride_h, const float *t_ramp, const float *f_ramp,
                               float *out, grafp_stream_t stream);
int grafp_peak_extract_bwd_f32(const float *spec, int B, int H, int W, int F, int KH, int KW, int stride_h,
                               const float *t_ramp, const float *f_ramp, const float *out, const float *grad_out,
                               float *dweight, float *dbias, void *workspace, size_t workspace_bytes,
                               grafp_stream_t stream);
size_t grafp_peak_extract_bwd_workspace(int B, int F, int KH, int KW);

/* ---- K3-K5: dynamic k-NN graph ---------------------------------------------------------------
 * Replaces DenseDilatedKnnGraph.forward (encoder/gcn_lib/torch_edge.py:270-284, y=None, dilation 1):
 * channel L2-normalise (:281) -> pairwise_distance (:7-18) -> topk(-dist,k) (:100).  The N x N matrix
 * is never materialised (exact-f32 MFMA tiles + register top-k).  Arithmetic order is the one fixed
 * in oracle/csrc/knn_graph.c; ties -> lowest index.  Centre indices (edge_index[1]) are arange(N)
 * and are not written.
 *   x (B,C,N) f32   idx (B,N,k) int64   1 <= k <= min(N, GRAFP_KNN_MAX_K)
 *   normalize != 0 applies the L2 normalisation (0 = dense_knn_matrix alone). */
size_t grafp_knn_graph_workspace(int B, int C, int N);
/* The two passes of grafp_knn_graph_f32, exposed so a caller can time / reuse them separately:
 *   normalize: x (B,C,N) -> xn (B,C,N) unit columns (copy when normalize == 0), sq (B,N) = ||xn||^2
 *   topk:      xn, sq -> idx (B,N,k) */
int grafp_knn_normalize_f32(const float *x, int B, int C, int N, int normalize, float *xn, float *sq,
                            grafp_stream_t stream);
int grafp_knn_topk_f32(const float *xn, const float *sq, int B, int C, int N, int k, int64_t *idx,
                       grafp_stream_t stream);
/* topk writing the compact int32 edge format the graph kernels also accept (halves the index traffic of the
 * gather kernels, which re-read the edges once per channel slab) */
int grafp_knn_topk_i32(const float *xn, const float *sq, int B, int C, int N, int k, int32_t *idx,
                       grafp_stream_t stream);
/* normalize pass reading any (b,c)-strided view with N contiguous: element (b,c,n) at x + b*stride_b + c*stride_c + n
 * (elements of `dtype`), e.g. the GEMM-friendly (C,B,N) layout (stride_b = N, stride_c = B*N). */
int grafp_knn_normalize_strided(const void *x, int dtype, int64_t stride_b, int64_t stride_c, int B, int C, int N,
                                int normalize, float *xn, float *sq, grafp_stream_t stream);
int grafp_knn_graph_f32(const float *x, int B, int C, int N, int k, int normalize, int64_t *idx, void *ws,
                        size_t ws_bytes, grafp_stream_t stream);

/* The same graph (bit-identical indices) through a bf16 pre-filter: Gram tiles on the bf16 matrix cores with a
 * rigorous rounding margin select 4-8 candidates per node, whose exact f32 distances (same arithmetic order) decide.
 * Shapes: C % 64 == 0, N % 128 == 0, N <= 65536, k <= 4 (grafp_knn_pre_supported); x is any (b, c) strided view with
 * N contiguous, f32 or bf16; idx is (B,N,k) int64 or int32 (idx_is_i32). */
int grafp_knn_pre_supported(int C, int N, int k);
size_t grafp_knn_pre_workspace(int B, int C, int N);
int grafp_knn_graph_pre(const void *x, int dtype, int64_t stride_b, int64_t stride_c, int B, int C, int N, int k,
                        int normalize, void *idx, int idx_is_i32, void *ws, size_t ws_bytes, grafp_stream_t stream);

/* ---- K3-K5 through a split-bf16 Gram matrix with certified results (knn_split.hip) -------------------------------
 * The same (B, N, k) neighbour indices as grafp_knn_graph_f32 with normalize = 1, bit for bit, for the shapes
 * grafp_knn_split_supported accepts (C % 32 == 0, N % 128 == 0, N <= 4096, k <= 4): every normalised feature is split
 * into two bf16 halves, the Gram matrix runs on the bf16 matrix cores (three products per 16 channels), each query
 * keeps its k + 1 smallest approximate distances, and a query whose consecutive distances are closer than twice the
 * rigorous error bound of the approximation (near-ties, duplicates) is recomputed with the exact f32 arithmetic of
 * grafp_knn_topk_f32.  x: any (b, c)-strided view with N contiguous, f32 or bf16 (as grafp_knn_normalize_strided);
 * idx int64 or int32 (idx_is_i32); n_uncertified: NULL, or one device int that receives the number of queries that
 * took the exact path (diagnostics).
 * bf16 inputs take the RAW form: a bf16 feature is its own exact bf16 operand, so the Gram matrix of the UN-normalised
 * features needs one product per 16 channels and no planes, and the normalisation is applied behind the product
 * (g = G / (den_q den_j)); tighter bound, same certified tiers, same indices.  The _for entries take the input dtype
 * (the plain ones answer for f32 inputs). */
int grafp_knn_split_supported(int C, int N, int k);
int grafp_knn_split_preferred(int C, int N, int k);   /* supported AND measured faster than grafp_knn_topk_f32 (C <= 128) */
int grafp_knn_split_preferred_for(int dtype, int C, int N, int k);
size_t grafp_knn_split_workspace(int B, int C, int N);
size_t grafp_knn_split_workspace_for(int dtype, int B, int C, int N);
int grafp_knn_graph_split(const void *x, int dtype, int64_t stride_b, int64_t stride_c, int B, int C, int N, int k,
                          void *idx, int idx_is_i32, void *ws, size_t ws_bytes, int32_t *n_uncertified,
                          grafp_stream_t stream);

/* ---- K6-K7: edge gather + max-relative aggregation -------------------------------------------
 * Replaces batched_index_select x2 (encoder/gcn_lib/torch_nn.py:79-98) + max over k of (x_j - x_i)
 * + channel interleave (torch_vertex.py:21-32):
 *   out[b,2c,n] = x[b,c,n] ; out[b,2c+1,n] = max_k ( x[b,c,idx[b,n,k]] - x[b,c,n] )
 * Backward: dx[b,c,n] = g[b,2c,n] - g[b,2c+1,n] + sum over m whose arg-max neighbour (first maximum)
 * is n of g[b,2c+1,m].  Indices outside [0,N) are clamped.
 *   x (B,C,N) f32   idx (B,N,K) int64   out/grad_out (B,2C,N) f32   dx (B,C,N) f32 */
int grafp_mrconv_fwd_f32(const float *x, const int64_t *idx, int B, int C, int N, int K, float *out,
                         grafp_stream_t stream);
int grafp_mrconv_bwd_f32(const float *x, const int64_t *idx, const float *grad_out, int B, int C, int N, int K,
                         float *dx, grafp_stream_t stream);
/* Same, any (b,c)-strided views with N contiguous, f32 or bf16 elements (out/grad_out have 2C channels; dx uses
 * x's strides). */
int grafp_mrconv_fwd_strided(const void *x, int dtype, int64_t x_sb, int64_t x_sc, const int64_t *idx, int B, int C,
                             int N, int K, void *out, int64_t o_sb, int64_t o_sc, grafp_stream_t stream);
int grafp_mrconv_bwd_strided(const void *x, int dtype, int64_t x_sb, int64_t x_sc, const int64_t *idx,
                             const void *grad_out, int64_t g_sb, int64_t g_sc, int B, int C, int N, int K, void *dx,
                             grafp_stream_t stream);

int grafp_mrconv_fwd_strided_i32(const void *x, int dtype, int64_t x_sb, int64_t x_sc, const int32_t *idx, int B, int C,
                                 int N, int K, void *out, int64_t o_sb, int64_t o_sc, grafp_stream_t stream);
int grafp_mrconv_bwd_strided_i32(const void *x, int dtype, int64_t x_sb, int64_t x_sc, const int32_t *idx,
                                 const void *grad_out, int64_t g_sb, int64_t g_sc, int B, int C, int N, int K,
                                 void *dx, grafp_stream_t stream);
/* Training form: the forward pass also records WHICH neighbour won (the first maximum: the routing rule of the backward
 * pass above, comparison for comparison) as 2 bits per element -- arg (B, C, N / 4) bytes, element n in bits 2 (n % 4)
 * of byte n / 4 -- and the backward pass runs from that record: no x, no neighbour gather, the same dx bit for bit.
 * For the shapes grafp_mrconv_arg_supported accepts (K <= 4, N % 4 == 0, N <= 2048, strides multiples of 4 elements;
 * pointers 4-element aligned).  idx int64 or int32 (idx_is_i32); dx with its own strides (d_sb, d_sc). */
int grafp_mrconv_arg_supported(int dtype, int64_t x_sb, int64_t x_sc, int64_t o_sb, int64_t o_sc, int N, int K);
int grafp_mrconv_fwd_arg(const void *x, int dtype, int64_t x_sb, int64_t x_sc, const void *idx, int idx_is_i32, int B,
                         int C, int N, int K, void *out, int64_t o_sb, int64_t o_sc, uint8_t *arg, grafp_stream_t stream);
int grafp_mrconv_bwd_arg(const uint8_t *arg, int dtype, const void *idx, int idx_is_i32, const void *grad_out,
                         int64_t g_sb, int64_t g_sc, int B, int C, int N, int K, void *dx, int64_t d_sb, int64_t d_sc,
                         grafp_stream_t stream);

/* ---- K8/K9 glue: fused [conv bias] + BatchNorm + activation + residual on the (C, M = B*N) layout ----
 * Replaces the `+ bias` -> BatchNorm2d -> ReLU/LeakyReLU -> `+ shortcut` chains around every 1x1 convolution
 * (encoder/gcn_lib/torch_vertex.py:152-162,191-193; torch_nn.py:56-60; encoder/graph_encoder.py:52-55,60-66,131-133).
 *   z[c][m] = act( ((x[c][m] + pre_bias[c]) - mean[c]) * invstd[c] * gamma[c] + beta[c] ) + residual[c][m]
 * training != 0: mean/var are the batch statistics of row c (biased variance); running_mean/var (optional) are
 * updated with `momentum` (unbiased variance), as nn.BatchNorm2d does.  training == 0: running statistics.
 * x, residual (optional), out, dz, dx: (C, M) rows contiguous, elements of `dtype`; everything else f32.
 * act: 0 none, 1 ReLU, 2 LeakyReLU(slope).  save_mean / save_invstd (C, groups) feed the backward.
 * groups (1..8, dividing M): every row is `groups` equal column segments with SEPARATE batch statistics and shared
 * gamma/beta -- the two views of a contrastive batch side by side; running statistics advance once per group in
 * order, i.e. exactly what `groups` sequential BatchNorm calls do (simclr/simclr.py:35,43).
 * Backward returns dx, dgamma, dbeta; the residual's gradient is dz itself; the gradient of pre_bias is exactly
 * zero in training mode (it cancels in the normalisation) and sum_m dx in eval mode. */
size_t grafp_bn_workspace(int C, int64_t M);
int grafp_bn_fwd(const void *x, int dtype, int C, int64_t M, int groups, const float *pre_bias, const float *gamma,
                 const float *beta, const void *residual, int act, float slope, float eps, float momentum,
                 int training, float *running_mean, float *running_var, void *out, float *save_mean,
                 float *save_invstd, void *ws, size_t ws_bytes, grafp_stream_t stream);
int grafp_bn_bwd(const void *x, const void *dz, int dtype, int C, int64_t M, int groups, const float *pre_bias,
                 const float *gamma,
                 const float *beta, const float *save_mean, const float *save_invstd, int act, float slope,
                 int training, void *dx, float *dgamma, float *dbeta, float *dpre_bias /* (C) or NULL */, void *ws,
                 size_t ws_bytes, grafp_stream_t stream);
/* Single-pass forms of the two entries above (same arguments, results and reference call sites, plus `sync`):
 * in training mode each workgroup keeps its chunk of a row (a few thousand elements, bf16 left packed) in registers across a row-wide rendezvous,
 * so x (and dz) cross HBM once instead of twice -- forward 3 -> 2 passes, backward 5 -> 3.
 *   sync  grafp_bn_sync_bytes(C, M) bytes, 8-byte aligned, ALL ONES (0xff) on entry; the call leaves it all ones
 *         again (so one buffer, filled once, serves every call enqueued on the same stream).  NULL, eval mode, rows
 *         that are not 16-byte aligned multiples of the vector width, or more than 256 chunks per row select the
 *         two-pass kernels.
 *   spin_limit  polls after which a workgroup stops waiting for the other workgroups of its row and recomputes their
 *         partial sums itself (same bits, more time): negative = the default (4096, ~4 ms), 0 = never wait. */
size_t grafp_bn_sync_bytes(int C, int64_t M);
int grafp_bn_fwd_1pass(const void *x, int dtype, int C, int64_t M, int groups, const float *pre_bias,
                       const float *gamma, const float *beta, const void *residual, int act, float slope, float eps,
                       float momentum, int training, float *running_mean, float *running_var, void *out,
                       float *save_mean, float *save_invstd, void *ws, size_t ws_bytes, int32_t *sync, int spin_limit,
                       grafp_stream_t stream);
int grafp_bn_bwd_1pass(const void *x, const void *dz, int dtype, int C, int64_t M, int groups, const float *pre_bias,
                       const float *gamma, const float *beta, const float *save_mean, const float *save_invstd,
                       int act, float slope, int training, void *dx, float *dgamma, float *dbeta,
                       float *dpre_bias /* (C) or NULL */, void *ws, size_t ws_bytes, int32_t *sync, int spin_limit,
                       grafp_stream_t stream);

/* ---- IVF-PQ parity index: asymmetric-distance scan ----------------------------------------------------------------
 * The index of the published protocol, faiss.IndexIVFPQ(IndexFlatL2(d), d, 64, 64, 8) with nprobe = 20
 * (eval.py:65-69,122; faiss==1.7.2 itself is not vendored -- its published algorithm is restated).  For query i and
 * its probe slot s (list probe[i][s], < 0 = skip) every code of that list gets
 *   out_dist[i*row_stride + out_start[i][s] + j] = sum_m || (q_i - centroid)_m - codebook[m][code_j[m]] ||^2
 *   out_pos [i*row_stride + out_start[i][s] + j] = list_start[list] + j          (row of the list-ordered code array)
 * centroids (nlist, d) f32, codebooks (M, 256, d/M) f32, codes (n, M) uint8 in list order (16-byte aligned),
 * list_start (nlist + 1) int64, probe (nq, nprobe) int32, out_start (nq, nprobe) int64.  The caller selects the top-k
 * of each row and maps positions to ids (grafp_amd/ivfpq.py). */
int grafp_ivfpq_scan_f32(const float *q, int nq, int d, const float *centroids, int nlist, const float *codebooks,
                         int M, const uint8_t *codes, const int64_t *list_start, const int32_t *probe, int nprobe,
                         const int64_t *out_start, int64_t row_stride, float *out_dist, int32_t *out_pos,
                         grafp_stream_t stream);

/* ---- IVF-PQ parity index: training, encoding, probing and the fused search (round 4) ------------------------------------
 * faiss.IndexIVFPQ.train / add / search of eval.py:65-69,122,212-213,269-270 (test_fp.py:276 defaults to it).  All
 * arithmetic is fixed so that oracle/csrc/ivfpq.c restates it bit for bit:
 *   residual r = x[row] - base[base_idx[row]]   (base == NULL: r = x[row]);   G sub-spaces of d = D / G dims;
 *   dist(row, g, j) = fmaf chain over c ascending of (r[g d + c] - cent[g][j][c])^2, start 0;   argmin: lowest j on ties.
 * grafp_pq_assign_f32: out (n, G) int32 and/or out_u8 (n, G) uint8 (k <= 256) -- coarse assignment (G = 1, k = nlist) and
 * PQ encoding (G = M, k = 256, base = the coarse centroids, base_idx = the coarse assignment) alike.
 * grafp_kmeans_f32: seeded Lloyd iterations wholly on the device.  cent (G, k, d) <- the residuals of rows init_rows[0..k);
 * `niter` times: assignment as above; per cluster the sum of its residuals over rows in 1024-row chunks (row order inside
 * a chunk, f32 adds), chunk sums added in chunk order, centroid = sum / (float)count, a cluster without rows keeps its
 * centroid.  faiss's own k-means (initialisation, sub-sampling order) is not reproducible without faiss: the trained
 * quantisers are comparable with it only statistically; everything downstream of them is deterministic. */
int grafp_pq_assign_f32(const float *x, int64_t n, int D, int G, const float *base /* (nbase, D) or NULL */,
                        const int32_t *base_idx /* (n) or NULL */, const float *cent /* (G, k, D/G) */, int k,
                        int32_t *out /* (n, G) or NULL */, uint8_t *out_u8 /* (n, G) or NULL */, grafp_stream_t stream);
size_t grafp_kmeans_workspace(int64_t n, int D, int G, int k);
int grafp_kmeans_f32(const float *x, int64_t n, int D, int G, const float *base, const int32_t *base_idx,
                     const int64_t *init_rows /* (k) */, int k, int niter, float *cent /* (G, k, D/G) */, void *ws,
                     size_t ws_bytes, grafp_stream_t stream);
/* probe (nq, nprobe) int32: the nprobe nearest lists of every query by (dist, list id), dist = fmaf chain of
 * (q_c - centroid_c)^2.  */
int grafp_ivfpq_probe_f32(const float *q, int nq, int d, const float *centroids, int nlist, int nprobe, int32_t *probe,
                          grafp_stream_t stream);
/* The search proper: for every query the k <= GRAFP_SEARCH_MAX_K smallest asymmetric distances over the codes of its
 * probed lists, ordered by (distance, id), ids = ids[position in the list-ordered code array] (< 2^31); missing results
 * are (+inf, -1).  Estimates as in grafp_ivfpq_scan_f32 (sub-space terms added in m order); no scratch. */
int grafp_ivfpq_search_f32(const float *q, int nq, int d, const float *centroids, int nlist, const float *codebooks, int M,
                           const uint8_t *codes, const int64_t *list_start, const int64_t *ids, const int32_t *probe,
                           int nprobe, int k, float *out_dist, int64_t *out_ids, grafp_stream_t stream);

/* Test helper (no reference counterpart, no state): a kernel that merely occupies `blocks` x `threads` CU slots for
 * `clocks` shader cycles -- the BatchNorm rendezvous is tested next to it. */
int grafp_debug_occupy(int blocks, int threads, int64_t clocks, grafp_stream_t stream);

/* ---- K9 forward / data gradient: the 1x1 convolution itself as a streaming bf16 GEMM, BatchNorm folded in ----
 * y[r][m] = sum_k w[r][k] * f(x[k][m]) for every Conv2d(1x1) of the encoder (encoder/gcn_lib/torch_vertex.py:152-162,
 * torch_nn.py:56-60, encoder/graph_encoder.py:21-24,52-55) and, with w transposed by the caller, its data gradient
 * (autograd of the same lines).  w (R, K/groups) bf16 row-major (the Conv2d weight layout: group g owns rows
 * [g*R/groups, ...) and operand rows [g*K/groups, ...)), x (K, M) bf16, y (R, M) bf16, rows contiguous, 16-byte aligned.
 *   views      number of equal column segments with separate BatchNorm statistics (the views of a contrastive batch)
 *   pro_tab    NULL, or (K, views, 2) f32 (scale, shift): f(x) = act(x * scale + shift), i.e. the BatchNorm +
 *              activation of the layer that PRODUCED x, applied while the operand tile sits in LDS (its normalised
 *              output is then never written: torch_nn.py:59-62 -> torch_vertex.py:160; graph_encoder.py:53-54,62-64);
 *              pro_act: 0 none, 1 ReLU, 2 LeakyReLU(pro_slope)
 *   stats_part NULL, or (R, views, P, 3) f32 with P = grafp_conv1x1_gemm_partials(...): per output row and view the
 *              shifted sums of the ROUNDED outputs (sum(y - s), sum((y - s)^2), s), which grafp_bn_finalize turns into
 *              the batch statistics of the BatchNorm that follows (torch_nn.py:60; torch_vertex.py:153,161)
 * Shapes: rows per group a multiple of 32, K per group a multiple of 32, columns per view a multiple of 128
 * (grafp_conv1x1_gemm_supported); anything else is the caller's library GEMM. */
int grafp_conv1x1_gemm_supported(int R, int K, int groups, int64_t M, int views);
int grafp_conv1x1_gemm_partials(int R, int K, int groups, int64_t M, int views);
/* The launch plan grafp_conv1x1_gemm_bf16 uses for this shape (a pure function of the arguments; no reference
 * counterpart -- it lets a caller or a test see WHICH size-dependent plan a launch takes): info (host, 8 ints) =
 * {tile configuration, workgroups, column tiles per workgroup, partials per (row, view), resident-workgroup target,
 *  tile rows, tile columns, row tiles}. */
int grafp_conv1x1_gemm_plan(int R, int K, int groups, int64_t M, int views, int *info);
int grafp_conv1x1_gemm_bf16(const void *w, const void *x, int R, int K, int groups, int64_t M, int views,
                            const float *pro_tab, int pro_act, float pro_slope, void *y, float *stats_part,
                            grafp_stream_t stream);

/* Inference form of [Conv2d(1x1) -> BatchNorm2d (eval: running statistics) -> activation] in ONE kernel (fingerprint
 * generation, generate.py:34-57 / test_fp.py:87-158 run the model in eval mode): z = act(bf16(W x) * tab.x + tab.y) with
 * tab (R, views, 2) from grafp_bn_finalize(training = 0).  Bit-identical to grafp_conv1x1_gemm_bf16 followed by
 * grafp_bn_affine_bf16 (the same arithmetic on the same rounded product); y is neither written nor re-read. */
int grafp_conv1x1_gemm_affine_bf16(const void *w, const void *x, int R, int K, int groups, int64_t M, int views,
                                   const float *tab, int act, float slope, void *z, grafp_stream_t stream);

/* y = W [x1; x2]: the operand is the row-wise concatenation of two (K1, M) / (K2, M) bf16 tensors (never materialised),
 * W (R, K1 + K2) bf16.  Used for the data gradient of the first layer of a residual block,
 * dX = [W^T | I] [dY; dZ]: the shortcut's gradient dZ (autograd's accumulate of
 * /root/reference/encoder/gcn_lib/torch_vertex.py:187-194 and encoder/graph_encoder.py:59-66, `x = ... + shortcut`)
 * enters the product against an identity block and the sum is rounded once. */
int grafp_conv1x1_gemm_cat_bf16(const void *w, const void *x1, int K1, const void *x2, int K2, int R, int64_t M, void *y,
                                grafp_stream_t stream);
/* The f32 ("parity") mode's products on the bf16 matrix cores instead of the library's f32 GEMM
 * (/root/reference/encoder/gcn_lib/torch_nn.py:56-60, encoder/graph_encoder.py:52-55 at the reference's f32 precision,
 * train.py:174-177): grafp_split_bf16_planes writes hi = bf16(v) and lo = bf16(v - hi) of n f32 values (n % 8 == 0);
 * grafp_conv1x1_gemm_split_f32 takes w3 (R, 3K) bf16 = [Wh | Wh | Wl] and x_planes (2K, M) bf16 = [Xh; Xl] and writes
 * y (R, M) f32 = Wh Xh + Wh Xl + Wl Xh, every partial product exact, summed in f32 accumulators (the dropped Wl Xl term
 * and the 16-bit representation: ~2^-16 relative).  K % 32 == 0, R % 32 == 0, M % 128 == 0. */
int grafp_split_bf16_planes(const float *x, int64_t n, void *hi, void *lo, grafp_stream_t stream);
int grafp_conv1x1_gemm_split_f32(const void *w3, const void *x_planes, int R, int K, int64_t M, float *y,
                                 grafp_stream_t stream);

/* All 1x1-convolution weights of one training step in ONE launch: per layer the bf16 copy (forward GEMM operand) and
 * the per-group transposed bf16 copy (data-gradient operand; with ld_t > R/g the columns beyond R/g -- an identity
 * block written once by the caller -- are left alone).  `table`: n_layers x 8 int64 words {src f32 (G*Rg, Kg), dst
 * bf16 (G*Rg, Kg), dst_t bf16 (G*Kg, ld_t), Rg, Kg, G, ld_t, first tile}, Rg and Kg multiples of 32; `tile_entry`:
 * layer index of every 32 x 32 tile.  Replaces the per-layer weight.to(bfloat16) / .t().contiguous() launches that
 * stand in for /root/reference/encoder/gcn_lib/torch_nn.py:56-60 (Conv2d weights are used as they are there). */
int grafp_weights_prepare(const void *table, const int32_t *tile_entry, int n_tiles, grafp_stream_t stream);
/* BatchNorm2d statistics from the GEMM's partial sums (training != 0; nn.BatchNorm2d semantics as grafp_bn_fwd:
 * biased variance for the normalisation, unbiased for the running update, once per view in order) or from the running
 * statistics (training == 0, stats_part ignored).  (C, K, groups, M, views) are the arguments of the GEMM launch that
 * wrote stats_part.  Writes save_mean / save_invstd (C, views) for grafp_bn_bwd and tab (C, views, 2) =
 * (gamma * invstd, beta + (pre_bias - mean) * gamma * invstd): z = act(y * tab.x + tab.y). */
int grafp_bn_finalize(const float *stats_part, int C, int K, int groups, int64_t M, int views, const float *pre_bias,
                      const float *gamma, const float *beta, float eps, float momentum, int training,
                      float *running_mean, float *running_var, float *save_mean, float *save_invstd, float *tab,
                      grafp_stream_t stream);
/* out = act(y * tab.x + tab.y) + residual over bf16 (C, M) rows: the apply half of BatchNorm + activation + shortcut
 * (torch_vertex.py:191-193, graph_encoder.py:65) for outputs that ARE materialised; one read (+ residual), one write. */
int grafp_bn_affine_bf16(const void *y, int C, int64_t M, int views, const float *tab, const void *residual, int act,
                         float slope, void *out, grafp_stream_t stream);

/* grafp_bn_finalize (training) and grafp_bn_affine_bf16 in ONE launch: every workgroup combines the partial sums of its
 * (row, view) itself; the first workgroup of a view saves mean / invstd / tab, the first of a row advances the running
 * statistics.  Same arguments and results as the two calls; 1 <= views <= 4. */
int grafp_bn_finalize_affine_bf16(const void *y, const float *stats_part, int C, int K, int groups, int64_t M, int views,
                                  const float *pre_bias, const float *gamma, const float *beta, float eps,
                                  float momentum, float *running_mean, float *running_var, float *save_mean,
                                  float *save_invstd, float *tab, const void *residual, int act, float slope,
                                  void *out, grafp_stream_t stream);

/* ---- K9 backward: weight gradient of a 1x1 convolution on the (C, M) layout ------------------------------
 * dW[o][c] = sum_m grad_out[o][m] * x[c][m] for every 1x1 Conv2d of the encoder (torch_vertex.py:152-162,
 * torch_nn.py:56, graph_encoder.py:52-55,131): tiny output, contraction over M = B*N with both operands contiguous
 * along it -- a split-K streaming reduction on bf16 MFMA instead of a library GEMM.
 *   grad_out (Cout, M) bf16, x (Cin, M) bf16, rows contiguous, 16-byte aligned; dweight (Cout, Cin/groups) f32
 *   (the Conv2d weight layout; group g owns output rows [g*Cout/groups, (g+1)*Cout/groups) and the matching
 *   input rows). */
size_t grafp_conv1x1_wgrad_workspace(int Cout, int Cin, int groups, int64_t M);
int grafp_conv1x1_wgrad_bf16(const void *grad_out, const void *x, int Cout, int Cin, int groups, int64_t M,
                             float *dweight, void *ws, size_t ws_bytes, grafp_stream_t stream);
/* The same weight gradient when x is the RAW output of the previous convolution and the operand of this layer was
 * act(BatchNorm(x)) applied on the fly (grafp_conv1x1_gemm_bf16's pro_tab): the (Cin, views, 2) table is applied to the
 * x tile in LDS again, so the normalised activation is never materialised in either direction.  views: column segments
 * with their own table entries (a split-K slice never straddles two); pro_tab NULL = plain weight gradient. */
size_t grafp_conv1x1_wgrad_pro_workspace(int Cout, int Cin, int groups, int64_t M, int views);
/* The launch plan of the weight gradient for this shape (as grafp_conv1x1_gemm_plan): info (host, 8 ints) =
 * {tile configuration (0 T, 1 S, 2 L, 3 S32, 4 M32, 5 L32, 6 SG, 7 LG, 8 T128, 10 S128; -1 = the register-staged kernel
 *  of odd shapes),
 *  tile output rows, tile operand rows, split-K slices, output tiles, 0, 0, 0}. */
int grafp_conv1x1_wgrad_plan(int Cout, int Cin, int groups, int64_t M, int views, int *info);
int grafp_conv1x1_wgrad_pro_bf16(const void *grad_out, const void *x, int Cout, int Cin, int groups, int64_t M,
                                 int views, const float *pro_tab, int pro_act, float pro_slope, float *dweight,
                                 void *ws, size_t ws_bytes, grafp_stream_t stream);
/* The same with the tile configuration as an explicit per-call argument (tile = -1: the measured rule, what the
 * entries above use; 0 ... 7: T 64x64, S 128x128, L 256x256, S32, M32 256x128, L32 (64-byte row pieces), SG, LG (G
 * operand through registers); 8, 10: T128, S128 = T and S on 256-byte row pieces, for operand rows that lie megabytes
 * apart; 9: the register-staged split-K kernel).  The rule picks the wide configurations only at
 * sizes a test cannot afford for every shape, so the tests force each one on small cases through this entry. */
size_t grafp_conv1x1_wgrad_tile_workspace(int Cout, int Cin, int groups, int64_t M, int views, int tile);
int grafp_conv1x1_wgrad_tile_bf16(const void *grad_out, const void *x, int Cout, int Cin, int groups, int64_t M,
                                  int views, const float *pro_tab, int pro_act, float pro_slope, int tile,
                                  float *dweight, void *ws, size_t ws_bytes, grafp_stream_t stream);
/* The two halves of the above as separate calls, for a training step's 64 weight gradients: _partials_ runs only the
 * split-K kernel (n_slices receives the number of partial sums per output element, laid out (n_slices, Cout, Cin/groups)
 * in ws); grafp_wgrad_reduce_multi later reduces the partial sums of many layers in ONE launch (host arrays of n_entries
 * workspace pointers / slice counts / output sizes Cout*Cin/groups / output pointers; the table travels in the kernel
 * arguments, so the launch is graph-capturable).  Same summation tree: the gradients are bit-identical to _tile_bf16's.
 * (/root/reference/encoder/gcn_lib/torch_nn.py:56-60 under loss.backward(), train.py:78) */
int grafp_conv1x1_wgrad_partials_bf16(const void *grad_out, const void *x, int Cout, int Cin, int groups, int64_t M,
                                      int views, const float *pro_tab, int pro_act, float pro_slope, int tile, void *ws,
                                      size_t ws_bytes, int *n_slices, grafp_stream_t stream);
int grafp_wgrad_reduce_multi(const void *const *parts, const int *n_slices, const int64_t *n_out, float *const *outs,
                             int n_entries, grafp_stream_t stream);
/* ---- optimizer update (round 6) ----------------------------------------------------------------------------------
 * optimizer.step() of /root/reference/train.py:79 for torch.optim.Adam with its defaults (train.py:174: betas
 * (0.9, 0.999), eps 1e-8, no weight decay, no amsgrad) on ALL parameter tensors in ceil(n_entries / 64) launches (the
 * table travels in the kernel arguments: graph-capturable, no device-side state).  Host arrays of n_entries device
 * pointers: the f32 parameter, its f32 gradient, exp_avg, exp_avg_sq (updated in place) and the parameter's step counter
 * (a f32 scalar holding the count AFTER this update -- the caller bumps it first, as torch does); numels < 2^31.
 * lr_dev: the learning rate as a f32 scalar on the device, read when the kernel RUNS (a scheduler may update it between
 * replays of a captured step), or NULL to use `lr`.  Per element, in f32 and in this order (no contraction):
 *   m = m + (g - m)(1 - beta1);  v = beta2 v + (1 - beta2) g g;
 *   p = p - ((float)(lr / (1 - beta1^t)) m) / (sqrt(v) / (float)sqrt(1 - beta2^t) + eps)      (corrections in double) */
int grafp_adam_multi_f32(float *const *params, const float *const *grads, float *const *exp_avgs,
                         float *const *exp_avg_sqs, const float *const *steps, const int64_t *numels, int n_entries,
                         const float *lr_dev, double lr, double beta1, double beta2, double eps, grafp_stream_t stream);
/* f32 operands (the f32 "parity" mode of the step): the same split-K streaming reduction with each value split into
 * hi = bf16(v), lo = bf16(v - hi) on the way into LDS and three bf16 MFMAs per tile step (Gh Xh + Gh Xl + Gl Xh; the
 * dropped Gl Xl term and the 16-bit representation are ~2^-16 relative, f32 accumulation).  Same layouts as above
 * with f32 elements. */
size_t grafp_conv1x1_wgrad_f32_workspace(int Cout, int Cin, int groups, int64_t M);
int grafp_conv1x1_wgrad_f32(const float *grad_out, const float *x, int Cout, int Cin, int groups, int64_t M,
                            float *dweight, void *ws, size_t ws_bytes, grafp_stream_t stream);

/* ---- K12: NT-Xent loss, fused forward + backward ----------------------------------------------
 * Replaces ntxent_loss (simclr/ntxent.py:4-29; called train.py:71).  Rows of the similarity matrix
 * are the 2*B_all embeddings (view i then view j; the loss is invariant to the reference's
 * interleaved order), S = z z^T / tau, loss_r = logsumexp_{c != r} S_rc - S_{r,partner(r)}.  S is
 * computed in exact-f32 MFMA tiles and never written.  Data-parallel form: every rank passes the
 * ALL-GATHERED embeddings and its own pair range [row_begin, row_begin + n_local); it receives the
 * gradient of the GLOBAL mean loss w.r.t. its own rows (no backward collective is needed) and its
 * share of the loss.  Single GPU: row_begin = 0, n_local = B_all.
 *   zi_all, zj_all (B_all, D) f32, D % 32 == 0, D <= 128
 *   loss_partial   (grafp_ntxent_num_partials(n_local)) f32: sum(loss_partial) / (2*B_all) = the
 *                  local rows' share of the mean loss
 *   dzi, dzj       (n_local, D) f32 gradient of the mean loss (unit upstream gradient)
 *   ws             grafp_ntxent_workspace(B_all) bytes: per-row log-sum-exp and positive logit of ALL 2*B_all rows plus
 *                  the partial (max, sum, positive) triples of the first pass's candidate splits (fixed combination
 *                  order: deterministic) */
size_t grafp_ntxent_workspace(int B_all);
int grafp_ntxent_num_partials(int n_local);
int grafp_ntxent_fwd_bwd_f32(const float *zi_all, const float *zj_all, int B_all, int D, int row_begin,
                             int n_local, float tau, float *loss_partial, float *dzi, float *dzj, void *ws,
                             size_t ws_bytes, grafp_stream_t stream);

/* ---- K10 glue: taps of the stride-2 node convolution of Downsample (graph_encoder.py:16-28) -------------
 *   x (rows, N) f32/bf16 contiguous rows (rows = C*B)  ->  out (3, rows, n_out), n_out = (N-1)/2 + 1,
 *   out[t][r][j] = x[r][2j + t - 1], zero outside [0, N); _bwd is its transpose (grad_out (3, rows, n_out) -> dx). */
int grafp_stride2_taps_fwd(const void *x, int dtype, int64_t rows, int N, void *out, grafp_stream_t stream);
int grafp_stride2_taps_bwd(const void *grad_out, int dtype, int64_t rows, int N, void *dx, grafp_stream_t stream);

/* ---- K13: brute-force fingerprint search ------------------------------------------------------
 * Replaces faiss.IndexFlatL2 add/search as used at eval.py:54,212-213,269-270: exact squared-L2
 * top-k, ascending, ids = row index + id_base, ties -> lowest id, id -1 / dist +inf when fewer than
 * k rows exist.  Arithmetic order fixed in oracle/csrc/flat_search.c.
 *   grafp_row_sqnorm_f32: the `index.add` step -- per-row squared norms of the resident database.
 *   db (n,d) f32   db_sqnorm (n) f32   q (nq,d) f32   d == 128   1 <= k <= GRAFP_SEARCH_MAX_K
 *   out_dist (nq,k) f32   out_ids (nq,k) int64 */
int grafp_row_sqnorm_f32(const float *m, int64_t n, int d, float *out, grafp_stream_t stream);
size_t grafp_knn_search_workspace(int64_t n, int nq, int d, int k);
int grafp_knn_search_l2_f32(const float *db, const float *db_sqnorm, int64_t n, const float *q, int nq, int d,
                            int k, int64_t id_base, float *out_dist, int64_t *out_ids, void *ws,
                            size_t ws_bytes, grafp_stream_t stream);

/* Same results (bit-identical ids and distances) from a bf16 pre-filter: the scan runs on `db_bf16`, a
 * round-to-nearest-even bf16 copy of db (grafp_f32_to_bf16; n*128 bf16, 16-byte aligned), with a rigorous error
 * margin, and exact f32 distances are evaluated only for the few hundred rows per query that can still be among
 * the k best.  Halves the bytes streamed per pass and lifts large batches off the exact-f32 matrix rate. */
int grafp_f32_to_bf16(const float *src, int64_t n_elems, void *dst, grafp_stream_t stream);
size_t grafp_knn_search_pre_workspace(int64_t n, int nq, int d, int k);
int grafp_knn_search_l2_pre(const float *db, const void *db_bf16, const float *db_sqnorm, int64_t n, const float *q,
                            int nq, int d, int k, int64_t id_base, float *out_dist, int64_t *out_ids, void *ws,
                            size_t ws_bytes, grafp_stream_t stream);

/* Merge P partial result lists (e.g. one per database shard/GPU after an all-gather):
 *   part_dist (P,nq,k) f32, part_ids (P,nq,k) int64 (id < 0 = empty) -> out (nq,k), by (dist,id). */
int grafp_merge_topk(const float *part_dist, const int64_t *part_ids, int P, int nq, int k, float *out_dist,
                     int64_t *out_ids, grafp_stream_t stream);

/* ---- sequence-level rerank of the segment search results ----------------------------
 * Replaces the per-item Python loop of eval.py:262-290 after ONE batched segment search:
 *   index_rows (n,128) f32      the resident database = the reference's fake_recon_index (dummy_db then db)
 *   q_rows (n_qrows,128) f32    the searched query segments; topk_ids (n_qrows,k) int64 their search results
 *   item i = query rows [item_row[i], item_row[i] + item_len[i]), item_len[i] <= max_len <= 64, max_len*k <= 2048
 *   out_ids (n_items,top) int64 candidate start ids, best first (score descending, then lowest id), -1 padded
 *   out_scores (n_items,top) f32 mean_t <q[t], index[cid+t]> over the rows that exist; -inf padded
 * Arithmetic order fixed in oracle/csrc/seq_rerank.c. */
int grafp_seq_rerank_f32(const float *index_rows, int64_t n, const float *q_rows, int64_t n_qrows,
                         const int64_t *topk_ids, int k, const int64_t *item_row, const int *item_len, int n_items,
                         int max_len, int top, int64_t *out_ids, float *out_scores, grafp_stream_t stream);

/* Sharded form: index_rows holds global rows [row_base, row_base + n_rows) of an n-row index (a contiguous shard plus
 * a halo of the next shard's first max_len - 1 rows); only candidates with start id in [id_lo, id_hi) are scored.
 * The per-shard (out_ids, out_scores) lists are merged by (score descending, id ascending). */
int grafp_seq_rerank_shard_f32(const float *index_rows, int64_t n_rows, int64_t row_base, int64_t n, int64_t id_lo,
                               int64_t id_hi, const float *q_rows, int64_t n_qrows, const int64_t *topk_ids, int k,
                               const int64_t *item_row, const int *item_len, int n_items, int max_len, int top,
                               int64_t *out_ids, float *out_scores, grafp_stream_t stream);

/* ---- device-side augmentation --------------------------------------------------------
 * Replaces the torch_audiomentations (==0.11.1, requirements.txt:4; not vendored) transforms composed at
 * modules/transformations.py:25-48 and applied per clip on DataLoader workers (:67-75) or per track (:98-107).
 * The impulse responses / noise recordings are resident RAGGED f32 banks: one flat buffer, recording r = elements
 * [start[r], start[r] + len[r]) (int64 starts, int32 lengths; no padding to the longest file); each of the B signals
 * picks a recording by index (index < 0: the signal is copied unchanged -- the transform's probability p).
 *
 * grafp_ir_convolve_f32 -- ApplyImpulseResponse(compensate_for_propagation_delay=False):
 *   out[b][t] = sum_{l=0}^{min(t, len-1)} ir[l] * x[b][t-l],  t < T  (full convolution truncated to the input length)
 *   one fmaf chain per output in increasing l (order fixed in oracle/csrc/augment.c); out must not alias x.
 *   ir_index may be NULL (every signal uses recording 0).
 * grafp_mix_snr_f32 -- AddBackgroundNoise: n[t] = recording index[b] at (offset[b] + t) mod len (the library concatenates
 *   random pieces of the file up to T samples; here one circular read from a random offset),
 *   out = x + rms(x) / 10^(snr_db[b]/20) * n / (rms(n) + 1e-8), rms over the T samples.  out may alias x. */
int grafp_ir_convolve_f32(const float *x, int64_t x_stride, int B, int T, const float *ir_bank,
                          const int64_t *ir_start, int n_ir, const int32_t *ir_len, const int32_t *ir_index /* (B) or NULL */, float *out,
                          int64_t out_stride, grafp_stream_t stream);
size_t grafp_mix_snr_workspace(int B, int T);
int grafp_mix_snr_f32(const float *x, int64_t x_stride, int B, int T, const float *noise_bank,
                      const int64_t *noise_start, int n_noise, const int32_t *noise_len, const int32_t *noise_index, const int32_t *noise_offset,
                      const float *snr_db, float *out, int64_t out_stride, void *ws, size_t ws_bytes,
                      grafp_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* GRAFP_HIP_H */
