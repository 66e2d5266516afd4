// gfx950 kernels of EM::mask, the reference's `--advanceEM` variant
// (/root/reference/src/refinement/EM.cpp:261-503):
//
//   k_mask_init   :266-323  order-0 pass over every window (window 0 receives no factor, :295-305;
//                           the sums run in the reference's sequential fp32 order so that the
//                           responsibilities -- and with them the cut-off and the window lists --
//                           are bit-identical); with optimizeQ the reference re-estimates q after
//                           EVERY sequence (:321), a serial chain that one wavefront walks
//   k_mask_hist / k_mask_pick   :329-343  the value at rank size_t(float(count)*f) of the
//                           descending order, by a three-pass radix select on the float bit
//                           patterns (the reference sorts all N*LW1 values)
//   k_mask_bits   :345-356  membership of every window in the top-f set, one bit per r slot
//   k_mask_e      :395-433  E-step over the listed windows only (full-width products, the
//                           reference's pos_[n][LW1-ri] and r_[n][0] indexing as written)
//   k_mask_m      :452-461  order-K counts of the listed windows, 2^-40 fixed point in LDS
//
// One wavefront owns one sequence at a time; the sequence's (K+1)-mers are decoded into a per-wave
// LDS array once and addressed at random from there.  Only ~f of the windows do work per
// iteration, so none of this is on the headline path: simplicity over the last factor of two.

#include "common.h"

#include <algorithm>
#include <type_traits>

namespace bamm {
namespace {

__device__ __forceinline__ void wave_lds_sync() {
    // DS operations of one wave retire in order; this only stops the compiler from moving LDS
    // reads above the writes of other lanes
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ void acc_add(long long* cell, long long v) {      // as device_utils.h:acc_add
    (void)__hip_atomic_fetch_add(cell, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// as device_utils.h: a sequence's statistics rounded to the accumulator's units before they are summed (exact sums)
__device__ __forceinline__ double stat_round_llh(double x) { return (x + 402653184.0) - 402653184.0; }      // 1.5 * 2^28
__device__ __forceinline__ double stat_round_sumr(double x) { return (x + 6291456.0) - 6291456.0; }          // 1.5 * 2^22

__device__ __forceinline__ float mask_wave_sum(float x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
    return x;
}

__device__ __forceinline__ unsigned long long mask_to_fixed40(float r) {   // as kernels.hip:to_fixed40
    const float a = r * 256.0f;
    const float hi_f = floorf(a);
    const uint32_t hi = (uint32_t)hi_f;
    const uint32_t lo = (uint32_t)((a - hi_f) * 4294967296.0f);
    return ((unsigned long long)hi << 32) | lo;
}

// WG: the arrays of a wave in the launch's global scratch (sequences beyond the LDS plan: the reference has no limit,
// EM.cpp:261-503) -- and there the listed r-indices are 32 bits wide (any length); in LDS they are 16 (<= 65 535 positions)
template <bool WG>
struct WaveLds {
    using lst_t = typename std::conditional<WG, uint32_t, uint16_t>::type;
    uint32_t* y;        // [max_len]  kmer_ mod 4^(K+1) per position
    float* f;           // [max_len]
    lst_t* lst;         // [max_len]  listed r-indices, ascending
    uint32_t* mb;       // [max_len/32 + 2]  membership words covering this sequence's r slots
};

template <bool WG>
__device__ __forceinline__ WaveLds<WG> wave_lds(unsigned char* base, const MaskKernelArgs& a, uint64_t wave) {
    unsigned char* p = base + (size_t)wave * a.wave_bytes;
    const size_t n4 = ((size_t)a.max_len * 4u + 15u) & ~(size_t)15u, nl = ((size_t)a.max_len * sizeof(typename WaveLds<WG>::lst_t) + 15u) & ~(size_t)15u;
    WaveLds<WG> w;
    w.y = reinterpret_cast<uint32_t*>(p);
    w.f = reinterpret_cast<float*>(p + n4);
    w.lst = reinterpret_cast<typename WaveLds<WG>::lst_t*>(p + 2 * n4);
    w.mb = reinterpret_cast<uint32_t*>(p + 2 * n4 + nl);
    return w;
}

// the arrays of wave `wave` of this block: one region per wave of the grid in the global scratch, or in LDS behind the block's table
template <bool WG>
__device__ __forceinline__ WaveLds<WG> wave_arrays(unsigned char* lds_base, const MaskKernelArgs& a, uint32_t wave, uint32_t wpb) {
    if constexpr (WG) return wave_lds<true>(a.wave_scratch, a, (uint64_t)blockIdx.x * wpb + wave);
    else return wave_lds<false>(lds_base, a, wave);
}

// kmer_[p] mod Y for every position of one sequence (Sequence.cpp:35-41) into LDS
__device__ __forceinline__ void decode_to_lds(const SeqView& sv, uint32_t seq, uint32_t L, uint32_t Y, int lane,
                                              uint32_t* ybuf) {
    const uint32_t* wp = sv.words + sv.word_off[seq];
    for (uint32_t p = (uint32_t)lane; p < L; p += 64u) {
        const uint32_t wi = p >> 4;
        const uint32_t lo = wp[wi], hi = wi ? wp[wi - 1u] : 0u;
        ybuf[p] = __builtin_amdgcn_alignbit(hi, lo, 30u - 2u * (p & 15u)) & (Y - 1u);
    }
    wave_lds_sync();
    const uint64_t e0 = sv.exc_off[seq], e1 = sv.exc_off[seq + 1];
    for (uint64_t e = e0 + (uint64_t)lane; e < e1; e += 64u) {       // N randomisation (Sequence.cpp:38)
        const uint2 x = sv.exc[e];
        if (x.x < L) ybuf[x.x] = x.y;
    }
    wave_lds_sync();
}

// the membership words of this sequence's r slots, then the listed r-indices in ascending order
template <bool WG>
__device__ __forceinline__ uint32_t load_list(const MaskKernelArgs& a, uint64_t base, uint32_t L, uint32_t LW1, int lane,
                                              const WaveLds<WG>& w) {
    const uint64_t w0 = base >> 5;
    const uint32_t nw = (uint32_t)(((base + L - 1u) >> 5) - w0) + 1u;
    for (uint32_t i = (uint32_t)lane; i < nw; i += 64u) w.mb[i] = a.bits[w0 + i];
    wave_lds_sync();
    uint32_t cnt = 0;
    for (uint32_t i0 = 0; i0 < LW1; i0 += 64u) {
        const uint32_t ri = i0 + (uint32_t)lane;
        const uint64_t slot = base + ri;
        const bool in = ri < LW1 && ((w.mb[(uint32_t)((slot >> 5) - w0)] >> (slot & 31u)) & 1u);
        const unsigned long long bal = __ballot(in);
        const uint32_t before = __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
        if (in) w.lst[cnt + before] = (typename WaveLds<WG>::lst_t)ri;
        cnt += (uint32_t)__popcll(bal);
    }
    wave_lds_sync();
    return cnt;
}

template <bool WG>
__device__ __forceinline__ bool listed(const WaveLds<WG>& w, uint64_t base, uint32_t ri) {
    const uint64_t slot = base + ri;
    return (w.mb[(uint32_t)((slot >> 5) - (base >> 5))] >> (slot & 31u)) & 1u;
}

// acc + x[0] + x[1] + ... strictly left to right; four values per LDS round trip
__device__ __forceinline__ float sequential_sum(float acc, const float* x, uint32_t n) {
    uint32_t i = 0;
    for (; i + 4u <= n; i += 4u) {
        const float4 v = *reinterpret_cast<const float4*>(x + i);
        acc += v.x; acc += v.y; acc += v.z; acc += v.w;
    }
    for (; i < n; i++) acc += x[i];
    return acc;
}

// ---- EM.cpp:266-323 ---------------------------------------------------------------------------
template <bool SERIAL, bool WG>
__global__ void __launch_bounds__(256) k_mask_init(MaskKernelArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    float* s0 = reinterpret_cast<float*>(smem);                       // [W][4], EM.cpp:270-274
    const uint32_t W = a.W;
    for (uint32_t i = threadIdx.x; i < W * 4u; i += blockDim.x) {
        const uint32_t j = i >> 2, y = i & 3u;
        s0[i] = a.v0[y * W + j] / a.vbg0[y];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const uint32_t wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    const WaveLds<WG> w = wave_arrays<WG>(smem + a.table_bytes, a, wave, wpb);
    float q = *a.q;
    float N1 = 0.0f;                                                  // EM.cpp:507, running over the processed sequences
    for (uint32_t seq = blockIdx.x * wpb + wave; seq < a.sv.count; seq += gridDim.x * wpb) {
        if (a.sv.mask && !a.sv.mask[seq]) continue;
        const uint32_t L = a.sv.len[seq], LW1 = L - W + 1u;
        const uint64_t base = a.sv.pos_off[seq];
        decode_to_lds(a.sv, seq, L, a.Y, lane, w.y);
        const float pos_i = q / (float)LW1;                          // :288
        for (uint32_t i = (uint32_t)lane; i < LW1; i += 64u) {       // window start i <-> r index L-W-i
            float p = 1.0f;
            if (i >= 1u)                                             // :301-303: j < min(W, ij) never reaches window 0
                for (uint32_t j = 0; j < W; j++) p *= s0[j * 4u + (w.y[i + j] & 3u)];
            w.f[L - W - i] = p * pos_i;                              // :309
        }
        wave_lds_sync();
        float nf = 1.0f - q;                                         // :286
        nf = sequential_sum(nf, w.f, LW1);                           // :310, the reference's order (all lanes alike)
        wave_lds_sync();
        for (uint32_t ri = (uint32_t)lane; ri < L; ri += 64u) {
            const float val = ri < LW1 ? w.f[ri] / nf : 0.0f;        // :315
            a.r[base + ri] = val;
            if (SERIAL && ri < LW1) w.f[ri] = val;
        }
        if (SERIAL) {                                                // :321 optimize_q() inside the sequence loop
            if (lane == 0) a.q_seq[seq] = q;
            wave_lds_sync();
            N1 = sequential_sum(N1, w.f, LW1);                       // later sequences still hold zeros (calloc)
            q = (a.n_total - N1 + 1.0f) / (a.n_total + 2.0f);        // :515
            wave_lds_sync();
        }
    }
    if (SERIAL && threadIdx.x == 0 && blockIdx.x == 0) *a.q = q;
}

// ---- EM.cpp:329-343 as a radix select ---------------------------------------------------------
__global__ void __launch_bounds__(256) k_mask_hist(MaskKernelArgs a, int pass) {
    __shared__ uint32_t h[2048];
    __shared__ unsigned long long npos;
    for (uint32_t i = threadIdx.x; i < 2048u; i += blockDim.x) h[i] = 0u;
    if (threadIdx.x == 0) npos = 0ull;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const uint32_t wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    const uint32_t prefix = a.sel->prefix;
    for (uint32_t seq = blockIdx.x * wpb + wave; seq < a.sv.count; seq += gridDim.x * wpb) {
        if (a.sv.mask && !a.sv.mask[seq]) continue;
        const uint32_t LW1 = a.sv.len[seq] - a.W + 1u;
        const uint64_t base = a.sv.pos_off[seq];
        for (uint32_t ri = (uint32_t)lane; ri < LW1; ri += 64u) {
            const uint32_t b = __float_as_uint(a.r[base + ri]);
            if (pass == 0) atomicAdd(&h[b >> 21], 1u);
            else if (pass == 1) { if ((b >> 21) == prefix) atomicAdd(&h[(b >> 10) & 0x7ffu], 1u); }
            else { if ((b >> 10) == prefix) atomicAdd(&h[b & 0x3ffu], 1u); }
        }
        if (pass == 0 && lane == 0) atomicAdd(&npos, (unsigned long long)LW1);
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < 2048u; i += blockDim.x)
        if (h[i]) acc_add(&a.hist[i], (long long)h[i]);
    if (pass == 0 && threadIdx.x == 0 && npos) acc_add(&a.hist[2048], (long long)npos);
}

__global__ void k_mask_pick(MaskKernelArgs a, int pass, float f) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    MaskSelect* s = a.sel;
    if (pass == 0) {
        s->pos_count = (double)a.hist[2048];
        const float scaled = (float)(unsigned long long)s->pos_count * f;   // EM.cpp:343
        double rank = (double)(unsigned long long)scaled;
        if (rank >= s->pos_count) rank = s->pos_count - 1.0;                // reference: out-of-bounds read
        s->rank = rank;
    }
    const int bins = pass == 2 ? 1024 : 2048;
    double cum = 0.0;
    int pick = 0;
    for (int b = bins - 1; b >= 0; b--) {
        const double hb = (double)a.hist[b];
        if (cum + hb > s->rank) { pick = b; break; }
        cum += hb;
    }
    s->rank -= cum;
    s->prefix = pass == 0 ? (uint32_t)pick : (pass == 1 ? ((s->prefix << 11) | (uint32_t)pick) : ((s->prefix << 10) | (uint32_t)pick));
    if (pass == 2) s->cutoff = __uint_as_float(s->prefix);
    for (int b = 0; b <= 2048; b++) a.hist[b] = 0ll;
}

// ---- EM.cpp:345-356 ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_mask_bits(MaskKernelArgs a) {
    const int lane = threadIdx.x & 63;
    const uint32_t wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    const float cutoff = a.sel->cutoff;
    unsigned long long mine = 0ull;
    for (uint32_t seq = blockIdx.x * wpb + wave; seq < a.sv.count; seq += gridDim.x * wpb) {
        if (a.sv.mask && !a.sv.mask[seq]) continue;
        const uint32_t LW1 = a.sv.len[seq] - a.W + 1u;
        const uint64_t base = a.sv.pos_off[seq];
        for (uint32_t ri = (uint32_t)lane; ri < LW1; ri += 64u) {
            if (a.r[base + ri] >= cutoff) {
                const uint64_t slot = base + ri;
                atomicOr(&a.bits[slot >> 5], 1u << (slot & 31u));
                mine++;
            }
        }
    }
    if (mine) atomicAdd(&a.sel->listed, mine);
}

// ---- EM.cpp:395-433 ---------------------------------------------------------------------------
template <bool S_IN_LDS, bool WG>
__global__ void __launch_bounds__(256) k_mask_e(MaskKernelArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    const uint32_t W = a.W, Ys = a.Y + 1u;
    float* s_lds = reinterpret_cast<float*>(smem);
    if (S_IN_LDS) {
        for (uint32_t i = threadIdx.x; i < W * Ys; i += blockDim.x) s_lds[i] = a.s[i];
        __syncthreads();
    }
    const float* s = S_IN_LDS ? s_lds : a.s;
    __shared__ double stat[4][3];
    const int lane = threadIdx.x & 63;
    const uint32_t wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    const WaveLds<WG> w = wave_arrays<WG>(smem + a.table_bytes, a, wave, wpb);
    const float q = *a.q;
    double llh = 0.0, sum_r = 0.0, nseq = 0.0;
    for (uint32_t seq = blockIdx.x * wpb + wave; seq < a.sv.count; seq += gridDim.x * wpb) {
        if (a.sv.mask && !a.sv.mask[seq]) continue;
        const uint32_t L = a.sv.len[seq], LW1 = L - W + 1u;
        const uint64_t base = a.sv.pos_off[seq];
        decode_to_lds(a.sv, seq, L, a.Y, lane, w.y);
        const uint32_t cnt = load_list(a, base, L, LW1, lane, w);
        const float q_first = a.q_seq ? a.q_seq[seq] : q;            // what pos_[n][.] holds since the order-0 pass
        const float pos_now = q / (float)LW1, pos_first = q_first / (float)LW1;
        float part = 0.0f;
        for (uint32_t e = (uint32_t)lane; e < cnt; e += 64u) {
            const uint32_t ri = w.lst[e];
            const uint32_t start = L - W - ri;                       // :412
            float p = 1.0f;
            for (uint32_t j = 0; j < W; j++) p *= s[j * Ys + w.y[start + j]];
            // :416 reads pos_[n][LW1-ri]: slot LW1 is never written (calloc: 0); slots of listed
            // indices were refreshed at :404, the others keep the order-0 pass's value
            const float posv = ri == 0u ? 0.0f : (listed(w, base, LW1 - ri) ? pos_now : pos_first);
            p *= posv;
            w.f[e] = p;
            part += p;
        }
        const float nf = (1.0f - q) + mask_wave_sum(part);           // :399,417
        if (lane == 0 && !(cnt && w.lst[0] == 0u)) a.r[base] = a.r[base] / nf;   // :421 (a listed slot 0 holds 0)
        for (uint32_t e = (uint32_t)lane; e < cnt; e += 64u) a.r[base + w.lst[e]] = w.f[e] / nf;   // :422-424
        llh += stat_round_llh((double)logf(nf));                     // :432; rounded per sequence as in the EM kernels
        sum_r += stat_round_sumr((double)((nf - (1.0f - q)) / nf));
        nseq += 1.0;
        wave_lds_sync();
    }
    if (lane == 0) { stat[wave][0] = llh; stat[wave][1] = sum_r; stat[wave][2] = nseq; }
    __syncthreads();
    if (threadIdx.x < 3) {
        double t = 0.0;
        for (uint32_t i = 0; i < wpb; i++) t += stat[i][threadIdx.x];
        a.partial_stat[(size_t)blockIdx.x * 4 + threadIdx.x] = t;
    }
}

// ---- EM.cpp:452-461, columns [j0, j1) ---------------------------------------------------------
// DIRECT: one column of the count table does not fit the LDS (order 7 and up: 4^(K+1) cells of 8 bytes) -- the listed
// windows' addends go straight into the pass's accumulator (device-scope integer atomics on 65 536+ rows per column:
// the same sums; contention is what the size of the table makes of it), all W columns in one launch
template <bool WG, bool DIRECT>
__global__ void __launch_bounds__(256) k_mask_m(MaskKernelArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    const uint32_t W = a.W, Y = a.Y, j0 = a.j0, j1 = a.j1;
    unsigned long long* tab = reinterpret_cast<unsigned long long*>(smem);   // [j1-j0][Y]
    if constexpr (!DIRECT) {
        for (uint32_t i = threadIdx.x; i < (j1 - j0) * Y; i += blockDim.x) tab[i] = 0ull;
        __syncthreads();
    }
    const int lane = threadIdx.x & 63;
    const uint32_t wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    const WaveLds<WG> w = wave_arrays<WG>(smem + a.table_bytes, a, wave, wpb);
    for (uint32_t seq = blockIdx.x * wpb + wave; seq < a.sv.count; seq += gridDim.x * wpb) {
        if (a.sv.mask && !a.sv.mask[seq]) continue;
        const uint32_t L = a.sv.len[seq], LW1 = L - W + 1u;
        const uint64_t base = a.sv.pos_off[seq];
        decode_to_lds(a.sv, seq, L, Y, lane, w.y);
        const uint32_t cnt = load_list(a, base, L, LW1, lane, w);
        for (uint32_t e = (uint32_t)lane; e < cnt; e += 64u) {
            const uint32_t ri = w.lst[e];
            const unsigned long long fx = mask_to_fixed40(a.r[base + ri] * a.fix_scale);
            if (fx == 0ull) continue;
            const uint32_t start = L - W - ri;
            if constexpr (DIRECT) {
                for (uint32_t j = j0; j < j1; j++) acc_add(a.acc_direct + (size_t)w.y[start + j] * W + j, (long long)fx);   // ABI layout [y][j]
            } else {
                for (uint32_t j = j0; j < j1; j++) atomicAdd(&tab[(j - j0) * Y + w.y[start + j]], fx);
            }
        }
        wave_lds_sync();
    }
    if constexpr (!DIRECT) {
        __syncthreads();
        unsigned long long* out = a.partial_n + (size_t)blockIdx.x * W * Y;  // [j][y], as k_reduce_partials reads it
        for (uint32_t i = threadIdx.x; i < (j1 - j0) * Y; i += blockDim.x) out[(size_t)j0 * Y + i] = tab[i];
    }
}

}  // namespace

size_t mask_wave_bytes(uint32_t max_len, bool wide_lists) {
    const size_t n4 = ((size_t)max_len * 4 + 15) & ~(size_t)15, nl = ((size_t)max_len * (wide_lists ? 4 : 2) + 15) & ~(size_t)15;
    const size_t nb = (((size_t)max_len / 32 + 2) * 4 + 15) & ~(size_t)15;
    return 2 * n4 + nl + nb;
}

#define BAMM_MASK_LAUNCH(KERNEL, LDS, ...)                                                          \
    do {                                                                                            \
        if ((LDS) > 48 * 1024) {                                                                    \
            hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void*>(KERNEL),              \
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)(LDS)); \
            if (e_ != hipSuccess) { set_error("hipFuncSetAttribute: %s", hipGetErrorString(e_)); return BAMM_ERR_HIP; } \
        }                                                                                           \
        hipLaunchKernelGGL(KERNEL, dim3(blocks), dim3(threads), (LDS), st, __VA_ARGS__);            \
        BAMM_HIP(hipGetLastError());                                                                \
    } while (0)

// LDS of a launch: the block's table, and the waves' arrays unless they live in the global scratch
static size_t mask_lds(const MaskKernelArgs& a, uint32_t threads) {
    return a.table_bytes + (a.wave_scratch ? 0 : (size_t)(threads / 64u) * a.wave_bytes);
}

int launch_mask_init(const MaskKernelArgs& a, bool serial, uint32_t blocks, uint32_t threads, hipStream_t st) {
    const bool wg = a.wave_scratch != nullptr;
    if (serial) {
        blocks = 1; threads = 64;
        if (wg) BAMM_MASK_LAUNCH((k_mask_init<true, true>), mask_lds(a, threads), a);
        else BAMM_MASK_LAUNCH((k_mask_init<true, false>), mask_lds(a, threads), a);
    } else {
        if (wg) BAMM_MASK_LAUNCH((k_mask_init<false, true>), mask_lds(a, threads), a);
        else BAMM_MASK_LAUNCH((k_mask_init<false, false>), mask_lds(a, threads), a);
    }
    return BAMM_OK;
}

int launch_mask_hist(const MaskKernelArgs& a, int pass, uint32_t blocks, hipStream_t st) {
    hipLaunchKernelGGL(k_mask_hist, dim3(blocks), dim3(256), 0, st, a, pass);
    BAMM_HIP(hipGetLastError());
    return BAMM_OK;
}

int launch_mask_pick(const MaskKernelArgs& a, int pass, float f, hipStream_t st) {
    hipLaunchKernelGGL(k_mask_pick, dim3(1), dim3(64), 0, st, a, pass, f);
    BAMM_HIP(hipGetLastError());
    return BAMM_OK;
}

int launch_mask_bits(const MaskKernelArgs& a, uint32_t blocks, hipStream_t st) {
    hipLaunchKernelGGL(k_mask_bits, dim3(blocks), dim3(256), 0, st, a);
    BAMM_HIP(hipGetLastError());
    return BAMM_OK;
}

int launch_mask_e(const MaskKernelArgs& a, bool s_in_lds, uint32_t blocks, uint32_t threads, hipStream_t st) {
    const size_t lds = mask_lds(a, threads);
    const bool wg = a.wave_scratch != nullptr;
    if (s_in_lds) { if (wg) BAMM_MASK_LAUNCH((k_mask_e<true, true>), lds, a); else BAMM_MASK_LAUNCH((k_mask_e<true, false>), lds, a); }
    else { if (wg) BAMM_MASK_LAUNCH((k_mask_e<false, true>), lds, a); else BAMM_MASK_LAUNCH((k_mask_e<false, false>), lds, a); }
    return BAMM_OK;
}

int launch_mask_m(const MaskKernelArgs& a, uint32_t blocks, uint32_t threads, hipStream_t st) {
    const size_t lds = mask_lds(a, threads);
    if (a.acc_direct != nullptr) {
        if (a.wave_scratch != nullptr) BAMM_MASK_LAUNCH((k_mask_m<true, true>), lds, a);
        else BAMM_MASK_LAUNCH((k_mask_m<false, true>), lds, a);
    } else {
        if (a.wave_scratch != nullptr) BAMM_MASK_LAUNCH((k_mask_m<true, false>), lds, a);
        else BAMM_MASK_LAUNCH((k_mask_m<false, false>), lds, a);
    }
    return BAMM_OK;
}

}  // namespace bamm
