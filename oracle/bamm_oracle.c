/* TEST INFRASTRUCTURE ONLY -- see bamm_oracle.h for the contract and parity status.
 *
 * Plain-C restatement of the BaMMmotif2 EM hot path.  "Faithful" means: the same loop order,
 * the same fp32 arithmetic, the same libc rand() protocol, OpenMP `parallel for` over
 * sequences with a compare-and-swap float add in the M-step -- so that with one thread the
 * results are bit-identical to the reference and with T threads they differ only by the
 * reference's own summation-order noise.
 *
 * Citations are /root/reference/src/<file>:<line>.
 */
#define _GNU_SOURCE
#include "bamm_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static size_t ipow4(size_t e) { return (size_t)1 << (2 * e); }

void orc_set_threads(int n) {
#ifdef _OPENMP
    omp_set_num_threads(n);
#else
    (void)n;
#endif
}

size_t orc_v_offset(size_t k, size_t W) { return W * ((ipow4(k + 1) - 4) / 3); }
size_t orc_v_size(size_t K, size_t W) { return orc_v_offset(K + 1, W); }
size_t orc_bg_offset(size_t k) { return (ipow4(k + 1) - 4) / 3; }
size_t orc_bg_size(size_t K) { return orc_bg_offset(K + 1); }

/* ---------------------------------------------------------------- sequence encoding -- */

size_t orc_seq_length(size_t L0, int single_strand) { return single_strand ? L0 : 2 * L0 + 1; }

/* Alphabet.cpp:46-55: complement of code c (1..4 -> 4..1).  Quirk kept on purpose: the
 * table entry for code 0 is the *character* 'N' (78), not 0, so an N in the forward strand
 * becomes the byte 78 in the reverse complement (and is then NOT randomised below). */
static uint8_t complement_code(uint8_t c) {
    if (c >= 1 && c <= 4) return (uint8_t)(5 - c);
    return (uint8_t)'N';
}

void orc_encode_sequence(const uint8_t* codes, size_t L0, int single_strand,
                         uint8_t* seq, uint64_t* kmer) {
    size_t L = orc_seq_length(L0, single_strand);
    memset(seq, 0, L);
    if (single_strand) {
        memcpy(seq, codes, L0);                       /* Sequence.cpp:15-17 */
    } else {
        for (size_t i = 0; i < L0; i++) {             /* Sequence.cpp:91-99 */
            seq[i] = codes[i];
            seq[2 * L0 - i] = complement_code(codes[i]);
        }                                             /* seq[L0] stays 0 = N separator */
    }
    /* Sequence.cpp:34-41: up to 11 digits, newest base = least significant digit, terms
     * visited from the oldest base to the newest, one rand() per term whose base is N. */
    for (size_t i = 0; i < L; i++) {
        uint64_t acc = 0;
        for (size_t k = (i < 10 ? i + 1 : 11); k > 0; k--) {
            uint8_t c = seq[i - k + 1];
            uint64_t digit = (c == 0) ? ((uint64_t)rand() % 4) : (uint64_t)(c - 1);
            acc += digit * ipow4(k - 1);
        }
        kmer[i] = acc;
    }
}

void orc_encode_set(const uint8_t* codes, const uint64_t* in_off, size_t N, int single_strand,
                    int do_srand, unsigned seed, uint8_t* seq_out, uint64_t* kmer_out,
                    uint64_t* out_off) {
    if (do_srand) srand(seed);                        /* mainBaMM.cpp:22 */
    uint64_t o = 0;
    for (size_t n = 0; n < N; n++) {
        size_t L0 = (size_t)(in_off[n + 1] - in_off[n]);
        out_off[n] = o;
        orc_encode_sequence(codes + in_off[n], L0, single_strand, seq_out + o, kmer_out + o);
        o += orc_seq_length(L0, single_strand);
    }
    out_off[N] = o;
}

/* ---------------------------------------------------------------- background model -- */

void orc_bg_model(const uint64_t* kmer, const uint64_t* off, size_t N, size_t K,
                  const float* alpha, float* v) {
    size_t total = orc_bg_size(K);
    uint64_t* cnt = (uint64_t*)calloc(total, sizeof(uint64_t));
    /* BackgroundModel.cpp:26-42: every position counts for every order (positions i<k see
     * implicit zero = 'A' padding in their upper digits). */
    for (size_t n = 0; n < N; n++)
        for (uint64_t i = off[n]; i < off[n + 1]; i++)
            for (size_t k = 0; k <= K; k++) cnt[orc_bg_offset(k) + kmer[i] % ipow4(k + 1)]++;

    /* BackgroundModel.cpp:441-473 (interpolate_ == true) */
    uint64_t base = 0;
    for (size_t y = 0; y < 4; y++) base += cnt[y];
    for (size_t y = 0; y < 4; y++)
        v[y] = ((float)cnt[y] + alpha[0] * 0.25f) / ((float)base + alpha[0]);
    for (size_t k = 1; k <= K; k++) {
        const uint64_t* nk = cnt + orc_bg_offset(k);
        const uint64_t* nk1 = cnt + orc_bg_offset(k - 1);
        float* vk = v + orc_bg_offset(k);
        const float* vk1 = v + orc_bg_offset(k - 1);
        for (size_t y = 0; y < ipow4(k + 1); y++) {
            size_t y2 = y % ipow4(k);   /* drop oldest base */
            size_t yk = y / 4;          /* drop newest base */
            vk[y] = ((float)nk[y] + alpha[k] * vk1[y2]) / ((float)nk1[yk] + alpha[k]);
        }
    }
    free(cnt);
}

/* ---------------------------------------------------------------- odds tables -- */

void orc_linear_s(const float* v, const float* vbg, size_t K, size_t W, size_t K_bg, float* s) {
    const float* vK = v + orc_v_offset(K, W);          /* Motif.cpp:485-494 */
    const float* b = vbg + orc_bg_offset(K_bg);
    size_t Y = ipow4(K + 1), Yb = ipow4(K_bg + 1);
    for (size_t y = 0; y < Y; y++)
        for (size_t j = 0; j < W; j++) s[y * W + j] = vK[y * W + j] / b[y % Yb];
}

void orc_log_s(const float* v, const float* vbg, size_t K, size_t W, size_t K_bg, float* s) {
    const float* vK = v + orc_v_offset(K, W);          /* Motif.cpp:471-483 */
    const float* b = vbg + orc_bg_offset(K_bg);
    size_t Y = ipow4(K + 1), Yb = ipow4(K_bg + 1);
    for (size_t y = 0; y < Y; y++)
        for (size_t j = 0; j < W; j++) s[y * W + j] = logf(vK[y * W + j] + 1e-5f) - logf(b[y % Yb]);
}

/* ---------------------------------------------------------------- E-step -- */

float orc_estep(const uint64_t* kmer, const uint64_t* off, size_t N, size_t K, size_t W,
                const float* s, float q, float* r_all) {
    float llikelihood = 0.0f;
    size_t Y = ipow4(K + 1);
#pragma omp parallel for reduction(+ : llikelihood)
    for (size_t n = 0; n < N; n++) {                  /* EM.cpp:148-196 */
        size_t L = (size_t)(off[n + 1] - off[n]);
        size_t LW1 = L - W + 1;
        const uint64_t* km = kmer + off[n];
        float* r = r_all + off[n];
        float normFactor = 1.0f - q;
        float pos_i = q / (float)LW1;
        for (size_t i = 0; i < LW1; i++) r[i] = 1.0f;
        for (size_t i = LW1; i < L; i++) r[i] = 0.0f; /* calloc / previous zeroing, EM.cpp:28,190 */
        /* EM.cpp:167-176: ij stops at LW1-1 (NOT L-1): trailing windows are truncated. */
        for (size_t ij = 0; ij < LW1; ij++) {
            size_t y = km[ij] % Y;
            for (size_t j = 0; j < W; j++) r[L - W - ij + j] *= s[y * W + j];
        }
        for (size_t i = 0; i < LW1; i++) {            /* EM.cpp:179-182 */
            r[i] *= pos_i;
            normFactor += r[i];
        }
        for (size_t i = 0; i < LW1; i++) r[i] /= normFactor;   /* EM.cpp:185-187 */
        for (size_t i = LW1; i < L; i++) r[i] = 0.0f;          /* EM.cpp:190-192 */
        llikelihood += logf(normFactor);                        /* EM.cpp:195 */
    }
    return llikelihood;
}

/* ---------------------------------------------------------------- M-step -- */

static inline void atomic_float_add(float* dst, float x) {   /* EM.cpp:203-215 */
    union { uint32_t u; float f; } old_v, new_v;
    do {
        old_v.f = *(volatile float*)dst;
        new_v.f = old_v.f + x;
    } while (!__atomic_compare_exchange_n((volatile uint32_t*)dst, &old_v.u, new_v.u, 0,
                                          __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST));
}

void orc_mstep_counts(const uint64_t* kmer, const uint64_t* off, size_t N, size_t K, size_t W,
                      const float* r_all, float* n) {
    size_t Y = ipow4(K + 1);
    memset(n, 0, orc_v_size(K, W) * sizeof(float));  /* EM.cpp:220-226 */
    float* nK = n + orc_v_offset(K, W);
#pragma omp parallel for
    for (size_t s_idx = 0; s_idx < N; s_idx++) {      /* EM.cpp:230-243 */
        size_t L = (size_t)(off[s_idx + 1] - off[s_idx]);
        const uint64_t* km = kmer + off[s_idx];
        const float* r = r_all + off[s_idx];
        for (size_t ij = 0; ij < L - W + 1; ij++) {
            size_t y = km[ij] % Y;
            for (size_t j = 0; j < W; j++) atomic_float_add(&nK[y * W + j], r[L - W - ij + j]);
        }
    }
    for (size_t k = K; k > 0; k--) {                  /* EM.cpp:247-254 */
        float* nk = n + orc_v_offset(k, W);
        float* nk1 = n + orc_v_offset(k - 1, W);
        for (size_t y = 0; y < ipow4(k + 1); y++) {
            size_t y2 = y % ipow4(k);
            for (size_t j = 0; j < W; j++) nk1[y2 * W + j] += nk[y * W + j];
        }
    }
}

void orc_update_v(const float* n, const float* A, const float* vbg, size_t K, size_t W, float* v) {
    /* Motif.h:95-136 */
    float* sumN = (float*)calloc(W, sizeof(float));
    for (size_t y = 0; y < 4; y++)
        for (size_t j = 0; j < W; j++) sumN[j] += n[y * W + j];
    for (size_t y = 0; y < 4; y++)
        for (size_t j = 0; j < W; j++)
            v[y * W + j] = (n[y * W + j] + A[j] * vbg[y]) / (sumN[j] + A[j]);
    for (size_t k = 1; k <= K; k++) {
        const float* nk = n + orc_v_offset(k, W);
        const float* nk1 = n + orc_v_offset(k - 1, W);
        float* vk = v + orc_v_offset(k, W);
        const float* vk1 = v + orc_v_offset(k - 1, W);
        const float* Ak = A + k * W;
        for (size_t y = 0; y < ipow4(k + 1); y++) {
            size_t y2 = y % ipow4(k);
            size_t yk = y / 4;
            for (size_t j = 0; j < k && j < W; j++) vk[y * W + j] = vk1[y2 * W + j];
            for (size_t j = k; j < W; j++)
                vk[y * W + j] = (nk[y * W + j] + Ak[j] * vk1[y2 * W + j]) / (nk1[yk * W + j - 1] + Ak[j]);
        }
    }
    free(sumN);
}

float orc_optimize_q(const float* r, const uint64_t* off, size_t N, size_t W) {
    float N1 = 0.f;                                   /* EM.cpp:505-519 */
    for (size_t n = 0; n < N; n++) {
        size_t L = (size_t)(off[n + 1] - off[n]);
        for (size_t i = 0; i < L - W + 1; i++) N1 += r[off[n] + i];
    }
    return ((float)N - N1 + 1.f) / ((float)N + 2.f);
}

void orc_calculate_p(const float* v, const float* vbg, size_t k_bg, size_t K, size_t W, float* p) {
    /* Motif.cpp:430-469 */
    for (size_t j = 0; j < W; j++)
        for (size_t y = 0; y < 4; y++) p[y * W + j] = v[y * W + j];
    for (size_t k = 1; k <= K; k++) {
        float* pk = p + orc_v_offset(k, W);
        const float* pk1 = p + orc_v_offset(k - 1, W);
        const float* vk = v + orc_v_offset(k, W);
        for (size_t y = 0; y < ipow4(k + 1); y++) {
            size_t yk = y / 4;
            for (size_t j = 0; j < k && j < W; j++) {
                float acc = 1;
                for (size_t i = 0; i <= j; i++) {
                    size_t yi = y / ipow4(i);
                    acc *= v[orc_v_offset(k - i, W) + yi * W + (j - i)];
                }
                for (size_t i = j + 1; i <= k; i++) {
                    if ((k - i) <= k_bg || k <= k_bg) {
                        size_t yi = y / ipow4(i);
                        acc *= vbg[orc_bg_offset(k - i) + yi];
                    } else {
                        size_t yi = y / 4 % ipow4(k_bg + 1);
                        acc *= vbg[orc_bg_offset(k_bg) + yi];
                    }
                }
                pk[y * W + j] = acc;
            }
            for (size_t j = k; j < W; j++) pk[y * W + j] = vk[y * W + j] * pk1[yk * W + j - 1];
        }
    }
}

/* ---------------------------------------------------------------- optimize loop -- */

size_t orc_optimize(const uint64_t* kmer, const uint64_t* off, size_t N, size_t K, size_t W,
                    size_t bg_order, const float* vbg, const float* A, float* v, float* q_io,
                    int optimizeQ, float epsilon, size_t max_iter, float* r, float* n,
                    float* trace_llh, float* trace_vdiff, float* llh_out) {
    size_t Y = ipow4(K + 1);
    size_t K_bg = bg_order < K ? bg_order : K;       /* EM.cpp:23 */
    float* s = (float*)malloc(Y * W * sizeof(float));
    float* v_before = (float*)malloc(Y * W * sizeof(float));
    float* vK = v + orc_v_offset(K, W);
    float q = *q_io;
    float llikelihood = 0.0f;                        /* EM.h:61 */
    int iterate = 1;
    size_t iteration = 0;
    while (iterate && iteration < max_iter) {        /* EM.cpp:81-128 */
        iteration++;
        float llikelihood_prev = llikelihood;
        memcpy(v_before, vK, Y * W * sizeof(float));
        orc_linear_s(v, vbg, K, W, K_bg, s);         /* EM.cpp:143 */
        llikelihood = orc_estep(kmer, off, N, K, W, s, q, r);
        orc_mstep_counts(kmer, off, N, K, W, r, n);
        orc_update_v(n, A, vbg, K, W, v);            /* EM.cpp:258 */
        if (optimizeQ && iteration <= 5) q = orc_optimize_q(r, off, N, W);   /* EM.cpp:99 */
        float v_diff = 0.0f;
        for (size_t y = 0; y < Y; y++)
            for (size_t j = 0; j < W; j++) v_diff += fabsf(vK[y * W + j] - v_before[y * W + j]);
        float llikelihood_diff = llikelihood - llikelihood_prev;
        if (trace_llh) trace_llh[iteration - 1] = llikelihood;
        if (trace_vdiff) trace_vdiff[iteration - 1] = v_diff;
        if (v_diff < epsilon) iterate = 0;                          /* EM.cpp:117 */
        if (llikelihood_diff < 0 && iteration > 10) iterate = 0;    /* EM.cpp:118 */
    }
    *q_io = q;
    if (llh_out) *llh_out = llikelihood;
    free(s);
    free(v_before);
    return iteration;
}

/* ---------------------------------------------------------------- masked EM (--advanceEM) -- */

static int cmp_float_desc(const void* a, const void* b) {
    float x = *(const float*)a, y = *(const float*)b;
    return (x < y) - (x > y);
}

size_t orc_mask(const uint64_t* kmer, const uint64_t* off, size_t N, size_t K, size_t W,
                size_t bg_order, const float* vbg, const float* A, float* v, float* q_io,
                int optimizeQ, float f, float epsilon, size_t max_iter, float* r, float* n,
                float* trace_llh, float* trace_vdiff, float* llh_out, float* cutoff_out,
                uint64_t* listed_out) {
    size_t Y = ipow4(K + 1);
    size_t K_bg = bg_order < K ? bg_order : K;
    float* s = (float*)calloc(Y * W, sizeof(float));
    float* pos = (float*)calloc((size_t)off[N] + 1, sizeof(float));    /* EM.cpp:26-32: calloc'ed [N][L] */
    float* v_before = (float*)malloc(Y * W * sizeof(float));
    float* vK = v + orc_v_offset(K, W);
    float q = *q_io;
    memset(r, 0, (size_t)off[N] * sizeof(float));

    /* EM.cpp:270-274: order-0 odds into the first four rows of s */
    for (size_t y = 0; y < 4; y++)
        for (size_t j = 0; j < W; j++) s[y * W + j] = v[y * W + j] / vbg[y];

    size_t pos_count = 0;
    for (size_t sq = 0; sq < N; sq++) {               /* EM.cpp:279-323 */
        size_t L = (size_t)(off[sq + 1] - off[sq]);
        size_t LW1 = L - W + 1;
        const uint64_t* km = kmer + off[sq];
        float* rn = r + off[sq];
        float* pn = pos + off[sq];
        float normFactor = 1.0f - q;
        float pos_i = q / (float)LW1;
        for (size_t i = 0; i < LW1; i++) { rn[i] = 1.0f; pn[i] = pos_i; }
        for (size_t ij = 0; ij < L; ij++) {           /* :295-305: window 0 never receives a factor */
            size_t y = km[ij] % 4;
            size_t padding = ((int)(ij - L + W) > 0) * (ij - L + W);
            for (size_t j = padding; j < (W < ij ? W : ij); j++) rn[L - W - ij + j] *= s[y * W + j];
        }
        for (size_t i = 0; i < LW1; i++) { rn[i] *= pn[L - W - i]; normFactor += rn[i]; }
        for (size_t i = 0; i < LW1; i++) rn[i] /= normFactor;
        if (optimizeQ) q = orc_optimize_q(r, off, N, W);   /* :321: inside the sequence loop */
        pos_count += LW1;
    }

    /* EM.cpp:329-343: full descending sort, cut-off at index size_t(float(count) * f) */
    float* r_all = (float*)malloc(pos_count * sizeof(float));
    size_t o = 0;
    for (size_t sq = 0; sq < N; sq++) {
        size_t LW1 = (size_t)(off[sq + 1] - off[sq]) - W + 1;
        memcpy(r_all + o, r + off[sq], LW1 * sizeof(float));
        o += LW1;
    }
    qsort(r_all, pos_count, sizeof(float), cmp_float_desc);
    float r_cutoff = r_all[(size_t)((float)pos_count * f)];
    free(r_all);
    if (cutoff_out) *cutoff_out = r_cutoff;

    /* EM.cpp:345-356: index lists (r-index, i.e. window start L-W-ri) */
    uint64_t* ri_off = (uint64_t*)calloc(N + 1, sizeof(uint64_t));
    for (size_t sq = 0; sq < N; sq++) {
        size_t LW1 = (size_t)(off[sq + 1] - off[sq]) - W + 1;
        uint64_t c = 0;
        for (size_t i = 0; i < LW1; i++) c += r[off[sq] + i] >= r_cutoff;
        ri_off[sq + 1] = ri_off[sq] + c;
    }
    uint32_t* ri = (uint32_t*)malloc((ri_off[N] + 1) * sizeof(uint32_t));
    for (size_t sq = 0; sq < N; sq++) {
        size_t LW1 = (size_t)(off[sq + 1] - off[sq]) - W + 1;
        uint64_t c = ri_off[sq];
        for (size_t i = 0; i < LW1; i++)
            if (r[off[sq] + i] >= r_cutoff) ri[c++] = (uint32_t)i;
    }
    if (listed_out) *listed_out = ri_off[N];

    float llikelihood_ = 0.0f;                        /* EM.h:61 */
    int iterate = 1;
    size_t iteration = 0;
    while (iterate && iteration < max_iter) {         /* EM.cpp:373-494 */
        iteration++;
        float llikelihood_prev = llikelihood_;
        memcpy(v_before, vK, Y * W * sizeof(float));
        float llikelihood = 0.0f;
        orc_linear_s(v, vbg, K, W, K_bg, s);
        for (size_t sq = 0; sq < N; sq++) {           /* :395-433 */
            size_t L = (size_t)(off[sq + 1] - off[sq]);
            size_t LW1 = L - W + 1;
            const uint64_t* km = kmer + off[sq];
            float* rn = r + off[sq];
            float* pn = pos + off[sq];
            const uint32_t* l = ri + ri_off[sq];
            size_t cnt = (size_t)(ri_off[sq + 1] - ri_off[sq]);
            float normFactor = 1.0f - q;
            float pos_i = q / (float)LW1;
            for (size_t idx = 0; idx < cnt; idx++) { rn[l[idx]] = 1.0f; pn[l[idx]] = pos_i; }
            for (size_t idx = 0; idx < cnt; idx++) {
                for (size_t j = 0; j < W; j++) {
                    size_t y = km[L - W - l[idx] + j] % Y;
                    rn[l[idx]] *= s[y * W + j];
                }
                rn[l[idx]] *= pn[LW1 - l[idx]];       /* :416: index LW1 (never written) for ri == 0 */
                normFactor += rn[l[idx]];
            }
            rn[0] /= normFactor;                      /* :421 */
            for (size_t idx = 0; idx < cnt; idx++) rn[l[idx]] /= normFactor;
            for (size_t i = LW1; i < L; i++) rn[i] = 0.0f;
            llikelihood += logf(normFactor);
        }
        llikelihood_ = llikelihood;
        memset(n, 0, orc_v_size(K, W) * sizeof(float));
        float* nK = n + orc_v_offset(K, W);
        for (size_t sq = 0; sq < N; sq++) {           /* :452-461, serial */
            size_t L = (size_t)(off[sq + 1] - off[sq]);
            const uint64_t* km = kmer + off[sq];
            const float* rn = r + off[sq];
            for (uint64_t c = ri_off[sq]; c < ri_off[sq + 1]; c++)
                for (size_t j = 0; j < W; j++) nK[(km[L - W - ri[c] + j] % Y) * W + j] += rn[ri[c]];
        }
        for (size_t k = K; k > 0; k--) {              /* :465-472 */
            float* nk = n + orc_v_offset(k, W);
            float* nk1 = n + orc_v_offset(k - 1, W);
            for (size_t y = 0; y < ipow4(k + 1); y++)
                for (size_t j = 0; j < W; j++) nk1[(y % ipow4(k)) * W + j] += nk[y * W + j];
        }
        orc_update_v(n, A, vbg, K, W, v);             /* :475 */
        float v_diff = 0.0f;
        for (size_t y = 0; y < Y; y++)
            for (size_t j = 0; j < W; j++) v_diff += fabsf(vK[y * W + j] - v_before[y * W + j]);
        float llikelihood_diff = llikelihood_ - llikelihood_prev;
        if (trace_llh) trace_llh[iteration - 1] = llikelihood_;
        if (trace_vdiff) trace_vdiff[iteration - 1] = v_diff;
        if (v_diff < epsilon) iterate = 0;                          /* :488 */
        if (llikelihood_diff < 0 && iteration > 10) iterate = 0;    /* :489 */
    }
    *q_io = q;
    if (llh_out) *llh_out = llikelihood_;
    free(s); free(pos); free(v_before); free(ri_off); free(ri);
    return iteration;
}

/* ---------------------------------------------------------------- scorer -- */

void orc_logodds(const uint64_t* kmer, const uint64_t* off, size_t N, size_t K, size_t W,
                 const float* s, float* mops, float* zoops, uint64_t* z) {
    size_t Y = ipow4(K + 1);
    size_t o = 0;
    for (size_t n = 0; n < N; n++) {                  /* ScoreSeqSet.cpp:41-66 */
        size_t L = (size_t)(off[n + 1] - off[n]);
        size_t LW1 = L - W + 1;
        const uint64_t* km = kmer + off[n];
        float maxScore = -FLT_MAX;
        size_t z_i = 0;
        for (size_t i = 0; i < LW1; i++) {
            float logOdds = 0.0f;
            for (size_t j = 0; j < W; j++) logOdds += s[(km[i + j] % Y) * W + j];
            mops[o++] = logOdds;
            if (logOdds > maxScore) { maxScore = logOdds; z_i = i; }
        }
        zoops[n] = maxScore;
        z[n] = z_i;
    }
}

/* ---------------------------------------------------------------- seeding from a PWM -- */

/* std::mt19937 (default seed 5489) restated from the published MT19937 recurrence. */
typedef struct { uint32_t mt[624]; int idx; } mt19937_t;
static void mt_seed(mt19937_t* g, uint32_t seed) {
    g->mt[0] = seed;
    for (int i = 1; i < 624; i++) g->mt[i] = 1812433253u * (g->mt[i - 1] ^ (g->mt[i - 1] >> 30)) + (uint32_t)i;
    g->idx = 624;
}
static uint32_t mt_next(mt19937_t* g) {
    if (g->idx >= 624) {
        for (int i = 0; i < 624; i++) {
            uint32_t yv = (g->mt[i] & 0x80000000u) | (g->mt[(i + 1) % 624] & 0x7fffffffu);
            uint32_t x = g->mt[(i + 397) % 624] ^ (yv >> 1);
            if (yv & 1u) x ^= 0x9908b0dfu;
            g->mt[i] = x;
        }
        g->idx = 0;
    }
    uint32_t y = g->mt[g->idx++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}
/* libstdc++ std::generate_canonical<double,53>(mt19937): two 32-bit draws, low word first. */
static double mt_canonical(mt19937_t* g) {
    double sum = 0.0, tmp = 1.0;
    for (int k = 0; k < 2; k++) { sum += (double)mt_next(g) * tmp; tmp *= 4294967296.0; }
    double ret = sum / tmp;
    if (ret >= 1.0) ret = nextafter(1.0, 0.0);
    return ret;
}

void orc_init_from_pwm(const float* pwm, size_t W, size_t K, const float* A, const float* vbg,
                       const uint64_t* kmer, const uint64_t* off, size_t N, float q, float* v) {
    orc_init_from_pwm_sites(pwm, W, K, A, vbg, kmer, off, N, q, v, NULL, NULL);
}

/* the same, also reporting the sampled site of every sequence (z_out[n]: 0 = no motif, i = window i-1; may be
 * NULL) and the integer site counts of all orders (counts_out[orc_v_size(K,W)], flat [k][y][j]; may be NULL) */
void orc_init_from_pwm_sites(const float* pwm, size_t W, size_t K, const float* A, const float* vbg,
                             const uint64_t* kmer, const uint64_t* off, size_t N, float q, float* v,
                             uint32_t* z_out, int* counts_out) {
    size_t total = orc_v_size(K, W);
    int* cnt = (int*)calloc(total, sizeof(int));
    /* Motif.cpp:205-220: floor at 1e-8 (double literal compare, float store), renormalise */
    for (size_t j = 0; j < W; j++) {
        float norm = 0.0f;
        for (size_t y = 0; y < 4; y++) {
            float x = pwm[y * W + j];
            v[y * W + j] = ((double)x <= 1.e-8) ? (float)1.e-8 : x;
            norm += v[y * W + j];
        }
        for (size_t y = 0; y < 4; y++) v[y * W + j] /= norm;
    }
    float* score = (float*)malloc(4 * W * sizeof(float));       /* Motif.cpp:229-233 */
    for (size_t y = 0; y < 4; y++)
        for (size_t j = 0; j < W; j++) score[y * W + j] = v[y * W + j] / vbg[y];

    mt19937_t rng;
    mt_seed(&rng, 5489u);                                       /* Motif.cpp:237 */
    size_t maxL = 0;
    for (size_t n = 0; n < N; n++) if (off[n + 1] - off[n] > maxL) maxL = (size_t)(off[n + 1] - off[n]);
    float* r = (float*)malloc((maxL + 2) * sizeof(float));
    double* cp = (double*)malloc((maxL + 2) * sizeof(double));
    for (size_t n = 0; n < N; n++) {                            /* Motif.cpp:255-311, serial */
        size_t L = (size_t)(off[n + 1] - off[n]);
        if (z_out) z_out[n] = 0;
        if (L < W) continue;                                    /* Motif.cpp:240-248 */
        size_t LW1 = L - W + 1;
        const uint64_t* km = kmer + off[n];
        float normFactor = 0.0f;
        float pos0 = 1.0f - q;
        float pos1 = q / (float)LW1;
        for (size_t i = 1; i <= LW1; i++) {
            r[i] = 1.0f;
            for (size_t j = 0; j < W; j++) r[i] *= score[(km[i - 1 + j] % 4) * W + j];
            r[i] *= pos1;
            normFactor += r[i];
        }
        r[0] = pos0;
        normFactor += r[0];
        for (size_t i = 0; i <= LW1; i++) r[i] /= normFactor;
        /* std::discrete_distribution (libstdc++ random.tcc): normalise in double, prefix
         * sums, last = 1.0, draw = lower_bound(cp, u) */
        size_t M = LW1 + 1;
        size_t z;
        if (M < 2) {
            z = 0;
        } else {
            double sum = 0.0;
            for (size_t i = 0; i < M; i++) sum += (double)r[i];
            double run = 0.0;
            for (size_t i = 0; i < M; i++) { run += (double)r[i] / sum; cp[i] = run; }
            cp[M - 1] = 1.0;
            double u = mt_canonical(&rng);
            size_t lo = 0, hi = M;                              /* lower_bound */
            while (lo < hi) { size_t mid = lo + (hi - lo) / 2; if (cp[mid] < u) lo = mid + 1; else hi = mid; }
            z = lo;
        }
        if (z_out) z_out[n] = (uint32_t)z;
        if (z > 0)                                              /* Motif.cpp:302-309 */
            for (size_t k = 0; k <= K; k++)
                for (size_t j = 0; j < W; j++)
                    cnt[orc_v_offset(k, W) + (km[z - 1 + j] % ipow4(k + 1)) * W + j]++;
    }
    for (size_t k = 1; k <= K; k++) {                           /* Motif.cpp:314-326 */
        const int* nk = cnt + orc_v_offset(k, W);
        const int* nk1 = cnt + orc_v_offset(k - 1, W);
        float* vk = v + orc_v_offset(k, W);
        const float* vk1 = v + orc_v_offset(k - 1, W);
        const float* Ak = A + k * W;
        for (size_t y = 0; y < ipow4(k + 1); y++) {
            size_t y2 = y % ipow4(k), yk = y / 4;
            for (size_t j = 0; j < k && j < W; j++) vk[y * W + j] = vk1[y2 * W + j];
            for (size_t j = k; j < W; j++)
                vk[y * W + j] = ((float)nk[y * W + j] + Ak[j] * vk1[y2 * W + j]) / ((float)nk1[yk * W + j - 1] + Ak[j]);
        }
    }
    if (counts_out) memcpy(counts_out, cnt, total * sizeof(int));
    free(cnt); free(score); free(r); free(cp);
}

/* ---------------------------------------------------------------- fp64 precision oracle -- */

void orc_em_step_f64(const uint64_t* kmer, const uint64_t* off, size_t N, size_t K, size_t W,
                     size_t bg_order, const float* vbg, const float* A, const float* v_in,
                     float q, float* v_out, float* n_out, double* llh_out, double* sum_r_out) {
    size_t Y = ipow4(K + 1);
    size_t K_bg = bg_order < K ? bg_order : K;
    size_t total = orc_v_size(K, W);
    double* s = (double*)malloc(Y * W * sizeof(double));
    double* n = (double*)calloc(total, sizeof(double));
    const float* vK = v_in + orc_v_offset(K, W);
    const float* b = vbg + orc_bg_offset(K_bg);
    for (size_t y = 0; y < Y; y++)
        for (size_t j = 0; j < W; j++) s[y * W + j] = (double)vK[y * W + j] / (double)b[y % ipow4(K_bg + 1)];
    double llh = 0.0, sum_r = 0.0;
    double* nK = n + orc_v_offset(K, W);
    size_t maxL = 0;
    for (size_t i = 0; i < N; i++) if (off[i + 1] - off[i] > maxL) maxL = (size_t)(off[i + 1] - off[i]);
    double* r = (double*)malloc(maxL * sizeof(double));
    for (size_t sidx = 0; sidx < N; sidx++) {
        size_t L = (size_t)(off[sidx + 1] - off[sidx]);
        size_t LW1 = L - W + 1;
        const uint64_t* km = kmer + off[sidx];
        double Z = 1.0 - (double)q;
        for (size_t i = 0; i < LW1; i++) {                       /* window start i */
            double p = 1.0;
            size_t cols = (LW1 - i < W) ? (LW1 - i) : W;        /* truncation, EM.cpp:167 */
            for (size_t j = 0; j < cols; j++) p *= s[(km[i + j] % Y) * W + j];
            r[i] = p * (double)q / (double)LW1;
            Z += r[i];
        }
        for (size_t i = 0; i < LW1; i++) { r[i] /= Z; sum_r += r[i]; }
        llh += log(Z);
        for (size_t ij = 0; ij < LW1; ij++) {
            size_t y = km[ij] % Y;
            size_t jmax = ij < W - 1 ? ij : W - 1;
            for (size_t j = 0; j <= jmax; j++) nK[y * W + j] += r[ij - j];
        }
    }
    for (size_t k = K; k > 0; k--) {
        double* nk = n + orc_v_offset(k, W);
        double* nk1 = n + orc_v_offset(k - 1, W);
        for (size_t y = 0; y < ipow4(k + 1); y++)
            for (size_t j = 0; j < W; j++) nk1[(y % ipow4(k)) * W + j] += nk[y * W + j];
    }
    double* v = (double*)malloc(total * sizeof(double));
    for (size_t j = 0; j < W; j++) {
        double sumN = 0;
        for (size_t y = 0; y < 4; y++) sumN += n[y * W + j];
        for (size_t y = 0; y < 4; y++) v[y * W + j] = (n[y * W + j] + (double)A[j] * (double)vbg[y]) / (sumN + (double)A[j]);
    }
    for (size_t k = 1; k <= K; k++) {
        const double* nk = n + orc_v_offset(k, W);
        const double* nk1 = n + orc_v_offset(k - 1, W);
        double* vk = v + orc_v_offset(k, W);
        const double* vk1 = v + orc_v_offset(k - 1, W);
        for (size_t y = 0; y < ipow4(k + 1); y++) {
            size_t y2 = y % ipow4(k), yk = y / 4;
            for (size_t j = 0; j < k && j < W; j++) vk[y * W + j] = vk1[y2 * W + j];
            for (size_t j = k; j < W; j++)
                vk[y * W + j] = (nk[y * W + j] + (double)A[k * W + j] * vk1[y2 * W + j]) / (nk1[yk * W + j - 1] + (double)A[k * W + j]);
        }
    }
    for (size_t i = 0; i < total; i++) { v_out[i] = (float)v[i]; if (n_out) n_out[i] = (float)n[i]; }
    if (llh_out) *llh_out = llh;
    if (sum_r_out) *sum_r_out = sum_r;
    free(s); free(n); free(r); free(v);
}
