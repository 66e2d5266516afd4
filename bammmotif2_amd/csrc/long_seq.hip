// Sequences beyond the register-resident kernels' reach (more than 8192 positions, reverse strand and
// separator included): the same E-step / M-step / scorer, one workgroup per sequence, window by window.
//
//   k_long_em     EM::EStep   refinement/EM.cpp:149-196, EM::MStep EM.cpp:231-243, the sum over r of
//                 EM::optimize_q EM.cpp:509-513, EM::getR's layout EM.cpp:173
//   k_long_score  ScoreSeqSet::calcLogOdds  seq_scoring/ScoreSeqSet.cpp:41-66
//
// The reference has no length limit (init/Sequence.cpp:4-43); the fast kernels hold a sequence in the
// registers of one wavefront.  Records this long are rare in motif discovery (ChIP-seq peaks are a few
// hundred bp), so this path is written for coverage, not speed: every window multiplies its W odds straight
// from the table in global memory (L2-resident, any order K), k-mers are read from the 2-bit stream with a
// binary search of the sequence's N-exception list, two passes (partition sum, then responsibilities), and
// the fixed-point addends go directly into the pass's global accumulator -- the same integers the fast
// kernels add, so mixing paths inside one set changes nothing.  Window products are multiplied left to right
// as in the reference (EM.cpp:167-176): bit-identical to it.
#include "device_utils.h"

#include <algorithm>
#include <cfloat>

namespace bamm {
namespace {

struct LongSeq {
    const uint32_t* words;       // the sequence's first word
    const uint2* exc;            // its exceptions (position, y), ascending
    uint32_t n_exc;
    uint32_t L;
};

__device__ __forceinline__ LongSeq open_seq(const SeqView& sv, uint32_t seq) {
    LongSeq s;
    s.words = sv.words + sv.word_off[seq];
    const uint64_t e0 = sv.exc_off[seq], e1 = sv.exc_off[seq + 1];
    s.exc = sv.exc + e0;
    s.n_exc = (uint32_t)(e1 - e0);
    s.L = sv.len[seq];
    return s;
}

// kmer_[p] mod Y (Y = 4^(K+1) <= 4^11): Sequence.cpp:35-41 from the 2-bit stream, or the exception's value
__device__ __forceinline__ uint32_t kmer_at(const LongSeq& s, uint32_t p, uint32_t Y) {
    uint32_t lo = 0, hi = s.n_exc;
    while (lo < hi) {                                      // first exception at or behind p
        const uint32_t mid = (lo + hi) >> 1;
        if (s.exc[mid].x < p) lo = mid + 1u; else hi = mid;
    }
    if (lo < s.n_exc && s.exc[lo].x == p) return s.exc[lo].y;
    const uint32_t wi = p >> 4;
    const uint32_t w_lo = s.words[wi], w_hi = wi ? s.words[wi - 1u] : 0u;
    return __builtin_amdgcn_alignbit(w_hi, w_lo, 30u - 2u * (p & 15u)) & (Y - 1u);
}

// product of window i (EM.cpp:167-176): columns j with i+j < LW1 only (the reference's loop bound)
__device__ __forceinline__ float window_product(const LongSeq& s, const float* tab, uint32_t i, uint32_t W, uint32_t Y,
                                                uint32_t LW1) {
    const uint32_t Ys = Y + 1u, cols = min(W, LW1 - i);
    float u = tab[kmer_at(s, i, Y)];                       // column 0: 1.0f * s == s
    for (uint32_t j = 1; j < cols; j++) u *= tab[(size_t)j * Ys + kmer_at(s, i + j, Y)];
    return u;
}

__device__ __forceinline__ double block_sum(double x, double* sh) {       // 256 threads, result in every thread
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63u) == 0u) sh[threadIdx.x >> 6] = x;
    __syncthreads();
    return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// slot_layout: r goes to state[p] with p = i + W - 1 (what the column-sliced E pass leaves) instead of the
// reference's r[L-W-i]
__global__ void __launch_bounds__(256) k_long_em(EmKernelArgs a, int accum, int write_r, int slot_layout) {
    if (a.stop != nullptr && *a.stop != 0u) return;      // optimize(): the stop rule fired in an earlier pass

    __shared__ double sh[4];
    const uint32_t W = a.W, Y = a.Y;
    const float q = *a.q, one_minus_q = 1.0f - q;
    double llh_acc = 0.0, sumr_acc = 0.0;
    uint32_t seq_cnt = 0;
    for (uint32_t t = blockIdx.x; t < a.sv.count; t += gridDim.x) {
        const uint32_t seq = pick_sequence(a.sv, t);
        if (write_r && !accum && a.seq_end && (seq < a.seq_begin || seq >= a.seq_end)) continue;
        if (a.sv.mask && !a.sv.mask[seq]) continue;
        const LongSeq s = open_seq(a.sv, seq);
        const uint32_t L = s.L, LW1 = L - W + 1u;
        const float pos_i = q / (float)LW1;                // EM.cpp:160
        double zpart = 0.0;
        for (uint32_t i = threadIdx.x; i < LW1; i += blockDim.x)
            zpart += (double)(window_product(s, a.s, i, W, Y, LW1) * pos_i);     // EM.cpp:180
        const float Z = one_minus_q + (float)block_sum(zpart, sh);               // EM.cpp:154,181
        const float invZ = 1.0f / Z;
        llh_acc += stat_round_llh((double)logf(Z));        // EM.cpp:195
        sumr_acc += 1.0 - stat_round_sumr((double)one_minus_q / (double)Z); // = sum_i r[i]  (EM.cpp:509-513)
        seq_cnt++;
        if (!accum && !write_r) continue;
        float* ro = write_r ? a.r_out + (a.sv.pos_off[seq] - a.r_base) : nullptr;
        for (uint32_t i = threadIdx.x; i < LW1; i += blockDim.x) {
            const float r = window_product(s, a.s, i, W, Y, LW1) * pos_i * invZ;   // EM.cpp:185-187
            if (write_r) ro[slot_layout ? i + W - 1u : L - W - i] = r;             // EM.cpp:173
            if (accum) {
                const unsigned long long F = to_fixed40(r * a.fix_scale);
                if (F != 0ull) {
                    const uint32_t cols = min(W, LW1 - i);                          // EM.cpp:236: ij < LW1
                    for (uint32_t j = 0; j < cols; j++)
                        acc_add(a.acc + (size_t)kmer_at(s, i + j, Y) * W + j, (long long)F);   // EM.cpp:240
                }
            }
        }
    }
    if (a.acc && threadIdx.x == 0) {
        acc_add_stat(a.acc, W * Y, 0, llh_acc);
        acc_add_stat(a.acc, W * Y, 1, sumr_acc);
        acc_add_stat(a.acc, W * Y, 2, (double)seq_cnt);
    }
}

__global__ void __launch_bounds__(256) k_long_score(ScoreKernelArgs a) {
    __shared__ float sbest[4];
    __shared__ uint32_t sidx[4];
    const uint32_t W = a.W, Y = a.Y, Ys = a.Y + 1u;
    for (uint32_t t = blockIdx.x; t < a.sv.count; t += gridDim.x) {
        const uint32_t seq = pick_sequence(a.sv, t);
        if (a.sv.mask && !a.sv.mask[seq]) continue;
        const LongSeq s = open_seq(a.sv, seq);
        const uint32_t LW1 = s.L - W + 1u;
        float* mo = a.mops ? a.mops + a.mops_off[seq] : nullptr;
        float best = -FLT_MAX;                              // ScoreSeqSet.cpp:46
        uint32_t best_i = 0;
        for (uint32_t i = threadIdx.x; i < LW1; i += blockDim.x) {
            float sc = 0.0f;                                // full windows, columns left to right (ScoreSeqSet.cpp:49-54)
            for (uint32_t j = 0; j < W; j++) sc += a.s[(size_t)j * Ys + kmer_at(s, i + j, Y)];
            if (mo) mo[i] = sc;
            if (sc > best) { best = sc; best_i = i; }       // ascending i per thread: the first maximum stays
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {                  // first arg-max over the wave, then over the four waves
            const float ob = __shfl_xor(best, o, 64);
            const uint32_t oi = __shfl_xor(best_i, o, 64);
            const bool take = (ob > best) || (ob == best && oi < best_i);
            best = take ? ob : best;
            best_i = take ? oi : best_i;
        }
        __syncthreads();
        if ((threadIdx.x & 63u) == 0u) { sbest[threadIdx.x >> 6] = best; sidx[threadIdx.x >> 6] = best_i; }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int w = 1; w < 4; w++)
                if (sbest[w] > best || (sbest[w] == best && sidx[w] < best_i)) { best = sbest[w]; best_i = sidx[w]; }
            a.zoops[seq] = best;
            a.z[seq] = best_i;
        }
    }
}

}  // namespace

int launch_long_em(const EmKernelArgs& a, bool accum, bool write_r, bool slot_layout, uint32_t blocks, hipStream_t st) {
    if (blocks == 0) return BAMM_OK;
    if (blocks == kPrimeOnly) return prime_kernel(reinterpret_cast<const void*>(&k_long_em));
    hipLaunchKernelGGL(k_long_em, dim3(blocks), dim3(256), 0, st, a, accum ? 1 : 0, write_r ? 1 : 0, slot_layout ? 1 : 0);
    BAMM_HIP(hipGetLastError());
    return BAMM_OK;
}

int launch_long_score(const ScoreKernelArgs& a, uint32_t blocks, hipStream_t st) {
    if (blocks == 0) return BAMM_OK;
    hipLaunchKernelGGL(k_long_score, dim3(blocks), dim3(256), 0, st, a);
    BAMM_HIP(hipGetLastError());
    return BAMM_OK;
}

}  // namespace bamm
