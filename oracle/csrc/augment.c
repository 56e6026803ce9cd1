/* oracle/csrc/augment.c -- CPU restatement of the two audio transforms of
 * /root/reference/modules/transformations.py:25-48.  TEST INFRASTRUCTURE (see oracle/__init__.py).
 *
 * The arithmetic lives in torch_audiomentations==0.11.1 (requirements.txt:4), which is neither vendored under
 * /root/reference nor installable here: PARITY UNPINNED against the library itself.  Restated from its published
 * definitions:
 *   ApplyImpulseResponse (compensate_for_propagation_delay=False): convolve(x, ir, mode="full")[..., :T]
 *   AddBackgroundNoise: background = rms_normalize(piece) = piece / (rms(piece) + 1e-8);
 *                       out = x + (rms(x) / 10^(snr_db/20)) * background
 * The convolution fixes an f32 order (one fmaf chain per output, taps in increasing order) that the HIP kernel
 * reproduces bit for bit; the mix uses double sums (the kernel's f32 tree sums agree to ~1e-6). */
#include <math.h>
#include <stdint.h>

void oracle_ir_convolve(const float *x, int B, int T, const float *ir_bank, const int64_t *ir_start, const int32_t *ir_len,
                        const int32_t *ir_index, float *out) {
#pragma omp parallel for schedule(dynamic, 64) collapse(2)
    for (int b = 0; b < B; ++b)
        for (int t = 0; t < T; ++t) {
            const int ii = ir_index ? ir_index[b] : 0;
            const float *xb = x + (int64_t)b * T;
            if (ii < 0) {
                out[(int64_t)b * T + t] = xb[t];
                continue;
            }
            const float *h = ir_bank + ir_start[ii];
            const int L = ir_len[ii];
            const int lmax = t < L - 1 ? t : L - 1;
            float acc = 0.0f;
            for (int l = 0; l <= lmax; ++l) acc = fmaf(h[l], xb[t - l], acc);
            out[(int64_t)b * T + t] = acc;
        }
}

void oracle_mix_snr(const float *x, int B, int T, const float *noise_bank, const int64_t *noise_start, const int32_t *noise_len,
                    const int32_t *noise_index, const int32_t *noise_offset, const float *snr_db, float *out) {
    for (int b = 0; b < B; ++b) {
        const float *xb = x + (int64_t)b * T;
        float *ob = out + (int64_t)b * T;
        const int ni = noise_index[b];
        if (ni < 0) {
            for (int t = 0; t < T; ++t) ob[t] = xb[t];
            continue;
        }
        const float *nb = noise_bank + noise_start[ni];
        const int nl = noise_len[ni];
        double sx = 0.0, sn = 0.0;
        for (int t = 0; t < T; ++t) {
            const double n = nb[((int64_t)noise_offset[b] + t) % nl];
            sx += (double)xb[t] * xb[t];
            sn += n * n;
        }
        const double rms_x = sqrt(sx / T), rms_n = sqrt(sn / T);
        const float scale = (float)((rms_x / pow(10.0, (double)snr_db[b] / 20.0)) / (rms_n + 1e-8));
        for (int t = 0; t < T; ++t) ob[t] = fmaf(scale, nb[((int64_t)noise_offset[b] + t) % nl], xb[t]);
    }
}
