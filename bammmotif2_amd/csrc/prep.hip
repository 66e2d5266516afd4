// Sequence::Sequence and BackgroundModel's counting pass on the device (see prep.h for the reference lines).
//
// What the host code of pack.cpp does sequence by sequence on every granted core -- reverse complement behind the N
// separator, kmer_[i] = sum over the last 11 bases of digit * 4^d with an unknown base replaced by ITS OWN rand() % 4 per
// (position, digit) term, the 2-bit stream of digit 0, and the list of positions whose 11-mer is not what the stream
// implies -- is position-parallel once two things are known per sequence: where its zero bytes are and where its draws
// start in the one rand() stream.  Both are prefix sums:
//   k_prep_count   per sequence: L, words, zero bytes (forward N and the separator), draws = sum min(11, L - z)
//   scans          word_off, pos_off, zero_off, draw_off               (k_scan_*: three small launches per array)
//   k_prep_zeros   the zero positions, ascending per sequence
//   [host]         the draws themselves: rand() % 4 of the stream srand(seed) starts (pack.cpp: rand_draws_mod4, jump-ahead
//                  on all host threads) -- the one piece that stays on the host, a 1-byte-per-draw upload
//   k_prep_pack    a lane per 32-bit word of the stream (16 positions), a wave per sequence: windows without a special byte
//                  are packed straight from the codes; next to one, kmer_[p] and the stream's own 11-mer are rebuilt term by
//                  term, every term's draw found by its index in the sequence's share of the stream:
//                      index(p, z) = draw_off + 11 * #{zeros < p - 10} + sum over zeros z' in [p - 10, p) of (p - z')
//                                    + rank of z among the zeros of [p - 10, p]
//                  (positions ascending, within a position the oldest base first: Sequence.cpp:35-41).  Run twice: once to
//                  count a sequence's exceptions (-> exc_off by a scan), once to write words and exceptions.
// Integer work throughout: the words and the exception list are the host path's, bit for bit (tests/test_prep_gpu.py).
//
// k_bg_counts: BackgroundModel.cpp:26-42 over a resident set: the (K+1)-mer of every position from the stream (a lane per
// word, the previous word supplies the context), overridden where the exception list holds the position; LDS histogram
// per block up to order 5, global atomics beyond.
#include "prep.h"

namespace bamm {
namespace {

constexpr uint32_t MASK22 = (1u << 22) - 1u;

__device__ __forceinline__ uint64_t wave_sum_u64(uint64_t x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
    return x;
}
__device__ __forceinline__ uint32_t wave_excl_scan_u32(uint32_t x, int lane, uint32_t* total) {
    uint32_t incl = x;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t y = __shfl_up(incl, o, 64);
        if (lane >= o) incl += y;
    }
    *total = __shfl(incl, 63, 64);
    return incl - x;
}

struct SeqGeom {                 // one record as the reference's Sequence sees it
    const uint8_t* c;
    uint64_t L0, L;
    bool ss;
    // the byte at position q of the (double-stranded) sequence: Sequence.cpp:18-32, Alphabet.cpp:46-55
    __device__ __forceinline__ uint32_t byte(uint64_t q) const {
        if (q < L0) return c[q];
        if (ss) return 0u;                                    // never asked for
        if (q == L0) return 0u;                               // the separator: code 0, randomised like any unknown base
        const uint32_t x = c[2 * L0 - q];
        return (x >= 1u && x <= 4u) ? 5u - x : 78u;           // the complement of code 0 is the BYTE 'N'
    }
};

__device__ __forceinline__ bool special(uint32_t b) { return b == 0u || b > 4u; }

__global__ void __launch_bounds__(256) k_prep_count(PrepArgs a) {
    const int lane = threadIdx.x & 63;
    const uint64_t waves = (uint64_t)gridDim.x * (blockDim.x >> 6);
    for (uint64_t n = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); n < a.n; n += waves) {
        const uint64_t o = a.off[n], L0 = a.off[n + 1] - o;
        const uint64_t L = a.single_strand ? L0 : 2 * L0 + 1;
        uint64_t nz = 0, nd = 0;
        for (uint64_t i = lane; i < ((L0 + 63) & ~uint64_t(63)); i += 64) {
            const bool z = i < L0 && a.codes[o + i] == 0;
            nz += z ? 1 : 0;
            nd += z ? (L - i < 11 ? L - i : 11) : 0;
        }
        nz = wave_sum_u64(nz);
        nd = wave_sum_u64(nd);
        if (!a.single_strand) { nz += 1; nd += (L - L0 < 11 ? L - L0 : 11); }
        if (lane == 0) {
            a.len[n] = (uint32_t)L;
            a.word_off[n + 1] = (L + 15) / 16;
            a.pos_off[n + 1] = L;
            a.zero_off[n + 1] = nz;
            a.draw_off[n + 1] = nd;
        }
    }
}

__global__ void __launch_bounds__(256) k_prep_zeros(PrepArgs a) {
    const int lane = threadIdx.x & 63;
    const uint64_t waves = (uint64_t)gridDim.x * (blockDim.x >> 6);
    for (uint64_t n = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); n < a.n; n += waves) {
        if (a.zero_off[n + 1] == a.zero_off[n]) continue;
        const uint64_t o = a.off[n], L0 = a.off[n + 1] - o;
        uint32_t* out = a.zero_pos + a.zero_off[n];
        uint32_t cnt = 0;
        for (uint64_t i0 = 0; i0 < L0; i0 += 64) {
            const uint64_t i = i0 + lane;
            const bool z = i < L0 && a.codes[o + i] == 0;
            const unsigned long long m = __ballot(z);
            if (z) out[cnt + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = (uint32_t)i;
            cnt += (uint32_t)__builtin_popcountll(m);
        }
        if (!a.single_strand && lane == 0) out[cnt] = (uint32_t)L0;
    }
}

// the zeros of one sequence and the draw bookkeeping that hangs on them
struct Zeros {
    const uint32_t* zp;
    uint32_t nz;
    const uint8_t* draws;        // the sequence's share of the stream starts here
    __device__ __forceinline__ uint32_t lower_bound(uint64_t x) const {       // first index with zp[i] >= x
        uint32_t lo = 0, hi = nz;
        while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (zp[mid] < x) lo = mid + 1; else hi = mid; }
        return lo;
    }
    // the draw of term (p, z): z is a zero position in [p - 10, p]
    __device__ __forceinline__ uint32_t draw(uint64_t p, uint64_t z) const {
        const uint32_t i0 = lower_bound(p >= 10 ? p - 10 : 0);
        uint64_t idx = 11ull * i0;
        uint32_t rank = 0;
        for (uint32_t i = i0; i < nz && zp[i] <= p; i++) {
            if (zp[i] < p) idx += p - zp[i];                   // the terms that zero had at the positions before p
            if (zp[i] < z) rank++;
        }
        return draws[idx + rank];
    }
};

template <bool WRITE>
__global__ void __launch_bounds__(256) k_prep_pack(PrepArgs a) {
    const int lane = threadIdx.x & 63;
    const uint64_t waves = (uint64_t)gridDim.x * (blockDim.x >> 6);
    for (uint64_t n = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); n < a.n; n += waves) {
        SeqGeom g;
        g.c = a.codes + a.off[n]; g.L0 = a.off[n + 1] - a.off[n]; g.ss = a.single_strand != 0;
        g.L = g.ss ? g.L0 : 2 * g.L0 + 1;
        Zeros zs;
        zs.zp = a.zero_pos + a.zero_off[n]; zs.nz = (uint32_t)(a.zero_off[n + 1] - a.zero_off[n]);
        zs.draws = a.draws + a.draw_off[n];
        const uint64_t nwords = (g.L + 15) / 16;
        uint32_t* words = WRITE ? a.words + a.word_off[n] : nullptr;
        uint64_t ebase = WRITE ? a.exc_off[n] : 0;             // exceptions written / counted so far
        for (uint64_t w0 = 0; w0 < nwords; w0 += 64) {
            const uint64_t w = w0 + lane;
            uint32_t word = 0, ec = 0;
            uint32_t epos[16], ekm[16], ecl[16];
            if (w < nwords) {
                const uint64_t p_first = 16 * w, p_last = (16 * w + 15 < g.L - 1) ? 16 * w + 15 : g.L - 1;
                // bytes [p_first - 10, p_last]: is there anything but A, C, G, T among them?
                bool any = false;
                uint32_t b[26];
#pragma unroll
                for (int i = 0; i < 26; i++) {
                    const int64_t q = (int64_t)p_first - 10 + i;
                    b[i] = (q >= 0 && (uint64_t)q <= p_last) ? g.byte((uint64_t)q) : 1u;
                    any = any || special(b[i]);
                }
                if (!any) {
#pragma unroll
                    for (int i = 0; i < 16; i++)
                        if (p_first + i <= p_last) word |= ((b[10 + i] - 1u) & 3u) << (30 - 2 * i);
                } else {
                    // base(q): digit 0 of kmer_[q], what the stream stores (an unknown base: the draw of ITS term (q, q))
                    auto base_at = [&](int i) -> uint32_t {    // i: index into b[], q = p_first - 10 + i
                        const uint64_t q = p_first - 10 + (uint64_t)i;
                        return (b[i] == 0u ? (uint32_t)zs.draw(q, q) : b[i] - 1u) & 3u;
                    };
                    uint32_t bs[26];
#pragma unroll
                    for (int i = 0; i < 26; i++) {
                        const int64_t q = (int64_t)p_first - 10 + i;
                        bs[i] = (q >= 0 && (uint64_t)q <= p_last) ? base_at(i) : 0u;
                    }
#pragma unroll
                    for (int i = 0; i < 16; i++) {
                        const uint64_t p = p_first + i;
                        if (p > p_last) continue;
                        word |= bs[10 + i] << (30 - 2 * i);
                        bool sp = false;
#pragma unroll
                        for (int e = 0; e <= 10; e++) sp = sp || ((int64_t)p - e >= 0 && special(b[10 + i - e]));
                        if (!sp) continue;                     // kmer_[p] is what the stream implies
                        uint32_t km = 0, clean = 0;
#pragma unroll
                        for (int e = 0; e <= 10; e++) {
                            if ((int64_t)p - e < 0) continue;
                            const uint32_t bb = b[10 + i - e];
                            const uint32_t digit = bb == 0u ? (uint32_t)zs.draw(p, p - e) : bb - 1u;   // its own draw per (position, digit)
                            km += digit << (2 * e);
                            clean |= bs[10 + i - e] << (2 * e);
                        }
                        km &= MASK22;
                        if (km != clean) { epos[ec] = (uint32_t)p; ekm[ec] = km; ecl[ec] = clean; ec++; }
                    }
                }
                if (WRITE) words[w] = word;
            }
            uint32_t total = 0;
            const uint32_t before = wave_excl_scan_u32(ec, lane, &total);
            if (WRITE) {
#pragma unroll
                for (int i = 0; i < 16; i++)
                    if ((uint32_t)i < ec) {
                        a.exc_pos[ebase + before + i] = epos[i];
                        a.exc_kmer[ebase + before + i] = ekm[i];
                        a.exc_clean[ebase + before + i] = ecl[i];
                    }
            }
            ebase += total;
        }
        if (!WRITE && lane == 0) a.exc_off[n + 1] = ebase;
    }
}

// ---- in-place inclusive scan of a u64 array (data[0] = 0 and data[i + 1] = count of item i on entry) ----------------------
constexpr uint32_t kScanThreads = 256, kScanMaxBlocks = 1024;

__global__ void __launch_bounds__(kScanThreads) k_scan_block(uint64_t* data, uint64_t n, uint64_t per_thread, uint64_t* block_sums, int phase) {
    __shared__ uint64_t sh[kScanThreads];
    const uint64_t begin = ((uint64_t)blockIdx.x * kScanThreads + threadIdx.x) * per_thread;
    const uint64_t end = begin + per_thread < n ? begin + per_thread : n;
    uint64_t s = 0;
    for (uint64_t i = begin; i < end; i++) s += data[i];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (uint32_t o = 1; o < kScanThreads; o <<= 1) {         // Hillis-Steele over the threads' sums
        const uint64_t y = threadIdx.x >= o ? sh[threadIdx.x - o] : 0;
        __syncthreads();
        sh[threadIdx.x] += y;
        __syncthreads();
    }
    if (phase == 0) {
        if (threadIdx.x == kScanThreads - 1) block_sums[blockIdx.x] = sh[threadIdx.x];
        return;
    }
    uint64_t run = (threadIdx.x ? sh[threadIdx.x - 1] : 0) + (blockIdx.x ? block_sums[blockIdx.x - 1] : 0);
    for (uint64_t i = begin; i < end; i++) { run += data[i]; data[i] = run; }
}

__global__ void __launch_bounds__(kScanMaxBlocks) k_scan_sums(uint64_t* block_sums, uint32_t blocks) {
    __shared__ uint64_t sh[kScanMaxBlocks];
    sh[threadIdx.x] = threadIdx.x < blocks ? block_sums[threadIdx.x] : 0;
    __syncthreads();
    for (uint32_t o = 1; o < kScanMaxBlocks; o <<= 1) {
        const uint64_t y = threadIdx.x >= o ? sh[threadIdx.x - o] : 0;
        __syncthreads();
        sh[threadIdx.x] += y;
        __syncthreads();
    }
    if (threadIdx.x < blocks) block_sums[threadIdx.x] = sh[threadIdx.x];
}

// ---- background counts --------------------------------------------------------------------------------------------------
template <bool LDS_HIST>
__global__ void __launch_bounds__(256) k_bg_counts(const uint32_t* words, const uint64_t* word_off, const uint32_t* len,
                                                   const uint64_t* exc_off, const uint2* exc, uint64_t n, uint32_t K,
                                                   unsigned long long* counts) {
    extern __shared__ uint32_t hist[];
    const uint32_t Y = 1u << (2 * (K + 1)), mask = Y - 1u;
    if (LDS_HIST) {
        for (uint32_t i = threadIdx.x; i < Y; i += blockDim.x) hist[i] = 0u;
        __syncthreads();
    }
    const int lane = threadIdx.x & 63;
    const uint64_t waves = (uint64_t)gridDim.x * (blockDim.x >> 6);
    for (uint64_t s = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); s < n; s += waves) {
        const uint32_t L = len[s];
        const uint32_t* w = words + word_off[s];
        const uint64_t e0 = exc_off[s], e1 = exc_off[s + 1];
        const uint32_t nwords = (L + 15u) / 16u;
        for (uint32_t wi = lane; wi < nwords; wi += 64) {
            const unsigned long long both = ((unsigned long long)(wi ? w[wi - 1] : 0u) << 32) | w[wi];
            // exceptions of this word: first one at or behind position 16 wi
            uint64_t lo = e0, hi = e1;
            while (lo < hi) { const uint64_t mid = (lo + hi) >> 1; if (exc[mid].x < 16u * wi) lo = mid + 1; else hi = mid; }
            for (uint32_t i = 0; i < 16u && 16u * wi + i < L; i++) {
                uint32_t y = (uint32_t)(both >> (30u - 2u * i)) & mask;
                if (lo < e1 && exc[lo].x == 16u * wi + i) { y = exc[lo].y & mask; lo++; }
                if (LDS_HIST) atomicAdd(&hist[y], 1u);
                else atomicAdd(&counts[y], 1ull);
            }
        }
    }
    if (LDS_HIST) {
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < Y; i += blockDim.x)
            if (hist[i]) atomicAdd(&counts[i], (unsigned long long)hist[i]);
    }
}

uint32_t prep_blocks(uint64_t n) {
    const uint64_t want = (n + 3) / 4;                        // four waves (sequences) per block
    return (uint32_t)(want < 1 ? 1 : (want > 8192 ? 8192 : want));
}

}  // namespace

int launch_prep_count(const PrepArgs& a, hipStream_t st) {
    if (a.n == 0) return BAMM_OK;
    hipLaunchKernelGGL(k_prep_count, dim3(prep_blocks(a.n)), dim3(256), 0, st, a);
    BAMM_HIP(hipGetLastError());
    return BAMM_OK;
}

int launch_prep_zeros(const PrepArgs& a, hipStream_t st) {
    if (a.n == 0) return BAMM_OK;
    hipLaunchKernelGGL(k_prep_zeros, dim3(prep_blocks(a.n)), dim3(256), 0, st, a);
    BAMM_HIP(hipGetLastError());
    return BAMM_OK;
}

int launch_prep_pack(const PrepArgs& a, bool write, hipStream_t st) {
    if (a.n == 0) return BAMM_OK;
    if (write) hipLaunchKernelGGL(k_prep_pack<true>, dim3(prep_blocks(a.n)), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(k_prep_pack<false>, dim3(prep_blocks(a.n)), dim3(256), 0, st, a);
    BAMM_HIP(hipGetLastError());
    return BAMM_OK;
}

int launch_scan_u64(uint64_t* data, uint64_t n, hipStream_t st) {
    if (n == 0) return BAMM_OK;
    uint64_t per_thread = (n + (uint64_t)kScanMaxBlocks * kScanThreads - 1) / ((uint64_t)kScanMaxBlocks * kScanThreads);
    if (per_thread < 4) per_thread = 4;
    const uint32_t blocks = (uint32_t)((n + per_thread * kScanThreads - 1) / (per_thread * kScanThreads));
    uint64_t* sums = nullptr;
    BAMM_HIP(hipMallocAsync((void**)&sums, kScanMaxBlocks * sizeof(uint64_t), st));
    hipLaunchKernelGGL(k_scan_block, dim3(blocks), dim3(kScanThreads), 0, st, data, n, per_thread, sums, 0);
    hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(kScanMaxBlocks), 0, st, sums, blocks);
    hipLaunchKernelGGL(k_scan_block, dim3(blocks), dim3(kScanThreads), 0, st, data, n, per_thread, sums, 1);
    BAMM_HIP(hipGetLastError());
    BAMM_HIP(hipFreeAsync(sums, st));
    return BAMM_OK;
}

int launch_bg_counts(const uint32_t* words, const uint64_t* word_off, const uint32_t* len, const uint64_t* exc_off,
                     const uint2* exc, uint64_t n, uint32_t K, unsigned long long* counts, uint32_t num_cus, hipStream_t st) {
    if (n == 0) return BAMM_OK;
    const uint32_t Y = 1u << (2 * (K + 1));
    // enough blocks that no block's LDS counter can wrap (a block sees at most ~n / blocks sequences of up to 2^32 positions:
    // the usual sets are far from it; sets of very long sequences take the global-atomics flavour)
    const uint32_t blocks = prep_blocks(n) < num_cus * 8u ? prep_blocks(n) : num_cus * 8u;
    if (K <= 5u) {
        hipLaunchKernelGGL(k_bg_counts<true>, dim3(blocks), dim3(256), Y * sizeof(uint32_t), st, words, word_off, len, exc_off, exc, n, K, counts);
    } else {
        hipLaunchKernelGGL(k_bg_counts<false>, dim3(blocks), dim3(256), 0, st, words, word_off, len, exc_off, exc, n, K, counts);
    }
    BAMM_HIP(hipGetLastError());
    return BAMM_OK;
}

}  // namespace bamm
