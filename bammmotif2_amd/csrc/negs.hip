// SeqGenerator's negative sampler on the device (/root/reference/src/seq_generator/SeqGenerator.cpp:63-348, restated on the
// host in bammmotif2_amd/host/fdr.cpp: kmer_frequency, rescale, draw).
//
// The reference draws m negatives per positive sequence, base by base, each base from the order-s conditionals of the whole
// set rescaled by the positive's own s-mer counts (:112-186), every base one rand() of libc's single stream (:222-341).  Which
// draw a base gets is known up front -- negative f of positive i consumes draws [d0[i] + f L, d0[i] + (f+1) L) -- so every
// negative can be sampled on its own once its generator state is known.  glibc's generator is linear (glibc_rand.h): the state
// after d draws is the seed state times t^d mod (t^31 - t^28 - 1), and t^d is the product of the host-made powers t^(2^b)
// over the set bits of d: a lane applies at most 48 of them to its 34-word state (each: 31 words of look-ahead, 34 x 31
// multiply-adds on registers), then steps through its L draws.
//   k_neg_counts   a wave per positive: its (k+1)-mer counts for k = 0..s from the resident set (the stream + the order-s
//                  exception list), added into the set's totals (:63-110; the host turns the totals into v and the bars)
//   k_neg_sample   a wave per positive: the counts again, lane 0 rescales the tables in the reference's float order into LDS
//                  (:112-186; --genericNeg: the set's own bars), then a lane per kept negative: state, draws, bases, 2-bit words
// Float expressions are the host restatement's, in its order (-ffp-contract=off, IEEE division): the negatives are the
// reference's, base for base (tests/test_negs_gpu.py against host/fdr.cpp::sample_negatives, which the CPU suite pins to the
// reference's own sequences).
#include "negs.h"

namespace bamm {
namespace {

__device__ __forceinline__ uint32_t bgoff(uint32_t k) { return ((1u << (2 * (k + 1))) - 4u) / 3u; }

// (k+1)-mer counts of one sequence, k = 0..s, into the wave's LDS table (zeroed by the caller): SeqGenerator.cpp:63-110
__device__ __forceinline__ void seq_counts(const NegArgs& a, uint64_t n, uint32_t* cnt, int lane) {
    const uint32_t L = a.len[n], s = a.s;
    const uint32_t* w = a.words + a.word_off[n];
    const uint64_t e0 = a.exc_off[n], e1 = a.exc_off[n + 1];
    const uint32_t nwords = (L + 15u) / 16u, mask = (1u << (2 * (s + 1))) - 1u;
    for (uint32_t wi = lane; wi < nwords; wi += 64) {
        const unsigned long long both = ((unsigned long long)(wi ? w[wi - 1] : 0u) << 32) | w[wi];
        uint64_t lo = e0, hi = e1;
        while (lo < hi) { const uint64_t mid = (lo + hi) >> 1; if (a.exc[mid].x < 16u * wi) lo = mid + 1; else hi = mid; }
        for (uint32_t i = 0; i < 16u && 16u * wi + i < L; i++) {
            const uint32_t j = 16u * wi + i;
            uint32_t y = (uint32_t)(both >> (30u - 2u * i)) & mask;
            if (lo < e1 && a.exc[lo].x == j) { y = a.exc[lo].y & mask; lo++; }
            for (uint32_t k = 0; k <= s; k++)
                if (j >= k) atomicAdd(&cnt[bgoff(k) + (y & ((1u << (2 * (k + 1))) - 1u))], 1u);
        }
    }
}

__global__ void __launch_bounds__(256) k_neg_counts(NegArgs a) {
    __shared__ uint32_t cnt_all[4][kNegTable];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t* cnt = cnt_all[wv];
    const uint32_t tot = bgoff(a.s + 1);
    const uint64_t waves = (uint64_t)gridDim.x * 4;
    for (uint64_t n = (uint64_t)blockIdx.x * 4 + wv; n < a.n; n += waves) {
        for (uint32_t i = lane; i < tot; i += 64) cnt[i] = 0u;
        __builtin_amdgcn_wave_barrier();
        seq_counts(a, n, cnt, lane);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (uint32_t i = lane; i < tot; i += 64)
            if (cnt[i]) atomicAdd(&a.total_counts[i], (unsigned long long)cnt[i]);
        __builtin_amdgcn_wave_barrier();
    }
}

// the state after the set bits of `d` were applied: r[0..33] = the next 34 outputs' history, oldest first
__device__ __forceinline__ void rand_jump(uint32_t (&r)[34], unsigned long long d, const uint32_t* pow2) {
    for (uint32_t b = 0; d != 0ull; b++, d >>= 1) {
        if (!(d & 1ull)) continue;
        const uint32_t* P = pow2 + 31u * b;                   // t^(2^b) mod (t^31 - t^28 - 1), uniform over the lanes
        uint32_t y[65];
#pragma unroll
        for (int k = 0; k < 34; k++) y[k] = r[k];
#pragma unroll
        for (int k = 34; k < 65; k++) y[k] = y[k - 3] + y[k - 31];
#pragma unroll
        for (int j = 0; j < 34; j++) {
            uint32_t v = 0;
#pragma unroll
            for (int c = 0; c < 31; c++) v += P[c] * y[c + j];
            r[j] = v;
        }
    }
}

__global__ void __launch_bounds__(256) k_neg_sample(NegArgs a) {
    __shared__ uint32_t cnt_all[4][kNegTable];
    __shared__ float bar_all[4][kNegTable];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t* cnt = cnt_all[wv];
    float* bar = bar_all[wv];
    const uint32_t s = a.s, tot = bgoff(s + 1);
    const uint64_t waves = (uint64_t)gridDim.x * 4;
    const uint64_t total_neg = a.n * a.m_fold;
    for (uint64_t n = (uint64_t)blockIdx.x * 4 + wv; n < a.n; n += waves) {
        const uint32_t L = a.len[n];
        if (a.generic) {
            for (uint32_t i = lane; i < tot; i += 64) bar[i] = a.bar[i];
        } else {
            for (uint32_t i = lane; i < tot; i += 64) cnt[i] = 0u;
            __builtin_amdgcn_wave_barrier();
            seq_counts(a, n, cnt, lane);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (lane == 0) {                                  // SeqGenerator.cpp:112-186, written for s = 2: the host's float order
                const float* v = a.v;
                const float A0 = a.A[0], A1 = a.A[1], A2 = a.A[2];
                float vs[kNegTable];
                float sum = 0.f;
                for (uint32_t y = 0; y < 4; y++) { vs[y] = v[y]; sum += vs[y]; bar[y] = sum; }
                {
                    const uint32_t o1 = 4, o0 = 0;
                    for (uint32_t y = 0; y < 16; y++) {
                        const uint32_t y2 = y % 4;
                        vs[o1 + y] = v[o1 + y] * ((float)cnt[o1 + y] + A0 * v[o0 + y2]) / v[o0 + y2] / ((float)L + A0);
                    }
                    float norm[4] = {0.f, 0.f, 0.f, 0.f};
                    for (uint32_t y = 0; y < 16; y++) {
                        const uint32_t yk = y / 4;
                        vs[o1 + y] = ((float)cnt[o1 + y] + A1 * vs[o1 + y]) / ((float)cnt[o0 + yk] + A1);
                        norm[yk] += vs[o1 + y];
                    }
                    for (uint32_t y = 0; y < 16; y++) vs[o1 + y] /= norm[y / 4];
                    for (uint32_t y = 0; y < 16; y++) {
                        if (y % 4 == 0) sum = 0.0f;
                        sum += vs[o1 + y];
                        bar[o1 + y] = sum;
                    }
                }
                {
                    const uint32_t o2 = 20, o1 = 4;
                    for (uint32_t y = 0; y < 64; y++) {
                        const uint32_t y2 = y % 16, yk = y / 4;
                        vs[o2 + y] = ((float)cnt[o2 + y] + A2 * vs[o1 + y2]) / ((float)cnt[o1 + yk] + A2);
                        if (y % 4 == 0) sum = 0.0f;
                        sum += vs[o2 + y];
                        bar[o2 + y] = sum;
                    }
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // this positive's kept negatives, a lane each
        const uint64_t first = n * a.m_fold;
        const uint32_t nwords = (L + 15u) / 16u;
        uint64_t kept_before = 0;                             // kept negatives of this positive in front of the current round
        for (uint64_t f0 = 0; f0 < a.m_fold; f0 += 64) {
            const uint64_t f = f0 + (uint64_t)lane, idx = first + f;
            const bool mine = f < a.m_fold && (a.keep_stride <= 1 || (idx % a.keep_stride == 0 && idx + a.keep_stride <= total_neg));
            const unsigned long long km = __ballot(mine);
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(km >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)km, 0u));
            if (mine) {
                uint32_t r[34];
#pragma unroll
                for (int k = 0; k < 34; k++) r[k] = a.seed_state[k];
                rand_jump(r, a.draw0[n] + f * (uint64_t)L, a.pow2);
                uint32_t* out = a.out_words + a.out_word_off[n] + (kept_before + rank) * (uint64_t)nwords;
                uint32_t word = 0, ctx = 0, pos = 0, bad = 0;
                const uint32_t ctx_mask = (1u << (2 * s)) - 1u;
                uint32_t seq_first[kNegMaxOrder];             // the first s bases (0-based codes) for the lower-order contexts
                while (pos < L) {
#pragma unroll
                    for (int j = 0; j < 34; j++) {
                        if (pos < L) {
                            const uint32_t v = r[(j + 3) % 34] + r[(j + 31) % 34];
                            r[j] = v;
                            const float random = (float)(int)(v >> 1) / 2147483647.0f;       // rand() / RAND_MAX in floats (:223)
                            uint32_t base;                    // 0..3
                            if (pos >= s) {
                                const float* b4 = bar + bgoff(s) + ctx * 4u;
                                base = (random > b4[0]) + (random > b4[1]) + (random > b4[2]);
                            } else if (pos == 0) {
                                base = random <= bar[0] ? 0u : (random <= bar[1] ? 1u : (random <= bar[2] ? 2u : (random <= bar[3] ? 3u : 4u)));
                                if (base == 4u) { bad = 1u; base = 0u; }      // the reference leaves seq[0] unset there (rand() == RAND_MAX)
                            } else {
                                uint32_t yk = 0;
                                for (uint32_t k = pos; k > 0; k--) yk += seq_first[pos - k] << (2 * k);
                                const float* b4 = bar + bgoff(pos) + yk;
                                base = (random > b4[0]) + (random > b4[1]) + (random > b4[2]);
                            }
                            if (pos < s) {
#pragma unroll
                                for (uint32_t q = 0; q < kNegMaxOrder; q++) if (q == pos) seq_first[q] = base;
                            }
                            ctx = ((ctx << 2) | base) & ctx_mask;
                            word |= base << (30u - 2u * (pos & 15u));
                            if ((pos & 15u) == 15u || pos + 1u == L) { out[pos >> 4] = word; word = 0; }
                            pos++;
                        }
                    }
                }
                if (bad) atomicAdd(a.bad, 1u);
            }
            kept_before += (uint64_t)__builtin_popcountll(km);
        }
        __builtin_amdgcn_wave_barrier();                      // the tables are rewritten for the next positive
    }
}

}  // namespace

int launch_neg_counts(const NegArgs& a, hipStream_t st) {
    if (a.n == 0) return BAMM_OK;
    const uint64_t want = (a.n + 3) / 4;
    hipLaunchKernelGGL(k_neg_counts, dim3((uint32_t)(want > 4096 ? 4096 : want)), dim3(256), 0, st, a);
    BAMM_HIP(hipGetLastError());
    return BAMM_OK;
}

int launch_neg_sample(const NegArgs& a, hipStream_t st) {
    if (a.n == 0) return BAMM_OK;
    const uint64_t want = (a.n + 3) / 4;
    hipLaunchKernelGGL(k_neg_sample, dim3((uint32_t)(want > 65535 ? 65535 : want)), dim3(256), 0, st, a);
    BAMM_HIP(hipGetLastError());
    return BAMM_OK;
}

}  // namespace bamm
