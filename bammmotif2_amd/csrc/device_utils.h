// Device-side helpers shared by the sequence kernels (kernels.hip, grouped.hip): DPP wave shifts,
// the 2^-40 fixed-point conversion, hand-issued LDS gathers / predicated LDS adds and the 2-bit
// sequence decode.  Everything is `__device__ __forceinline__` in an anonymous namespace.
#pragma once
#include "common.h"

namespace bamm {
namespace {

// ---- cross-lane helpers ------------------------------------------------------------------
// DPP wave shifts exist on the GFX9 family (incl. gfx950).  `oldv` is what lane 0 keeps.
__device__ __forceinline__ float wave_shr1(float oldv, float x) {
    return __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, oldv), __builtin_bit_cast(int, x),
                                           0x138, 0xf, 0xf, false));
}
// lane l takes lane l+1's value; lane 63 receives 0 (bound_ctrl: no pre-set destination, ONE v_mov_b32_dpp per
// dword -- with a destination to preserve the compiler emits a v_mov in front of every one of them)
__device__ __forceinline__ unsigned long long wave_shl1_u64(unsigned long long x) {
    const int lo = __builtin_amdgcn_mov_dpp((int)(unsigned)(x & 0xffffffffull), 0x130, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp((int)(unsigned)(x >> 32), 0x130, 0xf, 0xf, true);
    return ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo;
}
// f * (value of x in lane l-1), lane 0: f * 1 -- the E-chain's carry across lanes, as ONE v_mul_f32_dpp (a lane
// without a source keeps the destination, which holds f).  hipcc emits v_mov 1.0 / v_mov_dpp / v_mul for the
// same thing.  s_nop 1: a DPP read needs two wait states after a VALU write of its source, and the hazard
// recogniser does not look into inline asm.
__device__ __forceinline__ float mul_wave_shr1(float f, float x) {
    asm volatile("s_nop 1\n\tv_mul_f32_dpp %0, %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(f) : "v"(x));
    return f;
}
// r in [0,1] -> round(r * 2^40); exact integer accumulation up to 2^24 sequences per block
__device__ __forceinline__ unsigned long long to_fixed40(float r) {
    // exact power-of-two scalings and an exact split; the last unit (2^-40) is truncated
    const float a = r * 256.0f;
    const uint32_t hi = (uint32_t)a;                                           // truncation = floor for a >= 0
    const uint32_t lo = (uint32_t)(__builtin_amdgcn_fractf(a) * 4294967296.0f);   // v_fract_f32: a - floor(a), exact
    return ((unsigned long long)hi << 32) | lo;
}
// a * b with 0 * anything = 0 (v_mul_legacy_f32; identical to the IEEE product for non-zero finite operands)
__device__ __forceinline__ float mul_legacy(float a, float b) {
    float r;
    asm("v_mul_legacy_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// the same for a = r * 2^8 (the caller folds the power of two into a factor it applies anyway): same integer
__device__ __forceinline__ unsigned long long to_fixed40_pre(float a) {
    const uint32_t hi = (uint32_t)a;
    const uint32_t lo = (uint32_t)(__builtin_amdgcn_fractf(a) * 4294967296.0f);
    return ((unsigned long long)hi << 32) | lo;
}
constexpr double kFixedScaleInv = 1.0 / 1099511627776.0;
// the pass statistics travel in the same integer accumulator as the counts: log-likelihood in units of
// 2^-24 (|log Z| < 89 per sequence: 2^32 sequences fit), sum of r in units of 2^-30, the sequence count as is
constexpr double kLlhScale = 16777216.0, kSumrScale = 1073741824.0;
// no-return device-scope add (global_atomic_add_x2): the blocks of a pass fold their tables into one
__device__ __forceinline__ void acc_add(long long* cell, long long v) {
    (void)__hip_atomic_fetch_add(cell, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// The sequence count shares its word with a count of blocks whose statistics were not finite (log Z of a Z = 0 at
// q = 1, NaN odds from a degenerate model: the fp64 sums carry them, an integer cannot): bits 40.. of the word.  The
// update turns a non-zero count back into NaN for llh, which is what the reference's float sum would hold.
constexpr long long kStatBadUnit = 1ll << 40;
// A sequence's log-likelihood and sum of responsibilities are rounded to the accumulator's units (2^-24, 2^-30) BEFORE
// they are summed: fp64 sums of such values are exact (while below 2^29 / 2^23 per block: some 10^7 sequences), so a
// block's integer contribution is the exact sum of its sequences' and the int64 totals -- hence llh, q and every
// model that follows -- do not depend on how the sequences are split over waves, blocks or ranks.
// (x + C) - C with C = 1.5 * 2^(52 - bits): the first add rounds to nearest-even at 2^-bits, the second is exact.
__device__ __forceinline__ double stat_round_llh(double x) { return (x + 402653184.0) - 402653184.0; }      // 1.5 * 2^28
__device__ __forceinline__ double stat_round_sumr(double x) { return (x + 6291456.0) - 6291456.0; }          // 1.5 * 2^22
__device__ __forceinline__ double stat_nseq(long long word) { return (double)(word & (kStatBadUnit - 1ll)); }
__device__ __forceinline__ bool stat_bad(long long word) { return (word >> 40) != 0ll; }
// a block's three statistics (threads 0..2 call this with their own k)
__device__ __forceinline__ void acc_add_stat(long long* acc, uint32_t cells, uint32_t k, double v) {
    const double scaled = k == 0u ? v * kLlhScale : (k == 1u ? v * kSumrScale : v);
    if (k < 2u && !(fabs(scaled) < 9.0e18)) {              // inf, NaN, or beyond int64: flagged, not converted
        acc_add(acc + cells + 2u, kStatBadUnit);
        return;
    }
    acc_add(acc + cells + k, __double2ll_rn(scaled));
}
// a 64-bit value from another lane of the quad (DPP quad_perm control CTRL: 0xB1 = lanes 1,0,3,2; 0x4E = lanes 2,3,0,1)
template <int CTRL>
__device__ __forceinline__ unsigned long long quad_perm_u64(unsigned long long x) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_mov_dpp((int)(uint32_t)x, CTRL, 0xf, 0xf, true);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_mov_dpp((int)(uint32_t)(x >> 32), CTRL, 0xf, 0xf, true);
    return ((unsigned long long)hi << 32) | lo;
}
// 128-bit LDS gather.  hipcc splits a float4 LDS load whose components are consumed under
// different (even wave-uniform) conditions into b32/b64 pieces, which costs 2-4x the LDS cycles
// (tools/lds_bench2.hip), so the read is issued by hand; lds_wait() retires all of them and
// ties the results to the wait so that no consumer can be scheduled above it.
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));   // 16-byte global access at 4-byte alignment
__device__ __forceinline__ uint32_t lds_offset(const void* p) {
    return (uint32_t)(size_t)(const __attribute__((address_space(3))) void*)p;
}
// -DBAMM_PLAIN_LDS (tools/plain_lds_build.sh; a bisect build, never the shipped one): every hand-issued LDS instruction of
// the sequence kernels as the plain HIP statement it stands for -- loads the compiler sees, schedules and waits for itself,
// atomicAdd under an ordinary `if` -- so that a wrong result after a toolchain change can be pinned on (or cleared of) the
// hand-written part by running the same parity tests on both builds.  Slower (hipcc splits the 128-bit gathers); same results.
#ifdef BAMM_PLAIN_LDS
template <class T>
__device__ __forceinline__ T lds_load(uint32_t byte_addr) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef __attribute__((address_space(3))) const T lds_t;
    return *(lds_t*)(size_t)byte_addr;
#else
    return T{};
#endif
}
__device__ __forceinline__ uint32_t plain_lane() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
__device__ __forceinline__ void plain_lds_add(uint32_t byte_addr, unsigned long long v, unsigned long long mask) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef __attribute__((address_space(3))) unsigned long long lds_u64;
    if ((mask >> plain_lane()) & 1ull) atomicAdd((unsigned long long*)(lds_u64*)(size_t)byte_addr, v);
#endif
}
#endif
__device__ __forceinline__ f32x4 lds_read_b128(uint32_t byte_addr) {
#ifdef BAMM_PLAIN_LDS
    return lds_load<f32x4>(byte_addr);
#else
    f32x4 v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(byte_addr));
    return v;
#endif
}
// what was read by hand has landed (and no consumer moves above the wait); the plain build's loads are the compiler's to wait for
template <class T, int M>
__device__ __forceinline__ void lds_wait(T (&v)[M]) {
#ifndef BAMM_PLAIN_LDS
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]));
#pragma unroll
    for (int m = 1; m < M; m++) asm volatile("" : "+v"(v[m]) : "v"(v[0]));
#endif
}

// One predicated 64-bit LDS add: exec <- exec & pad & nz (both wave masks in SGPR pairs; the incoming
// exec is saved and restored, so a divergent caller keeps its masked-off lanes off), ds_add_u64.
// Replaces the compiler's v_cmp_u64 / s_nor / s_and_saveexec / s_or sequence per atomic.  The adds are
// fire-and-forget (no return): lds_drain() must run before anyone reads the table.
__device__ __forceinline__ void lds_add_u64_masked(uint32_t byte_addr, unsigned long long v,
                                                   unsigned long long pad_mask, unsigned long long nz_mask) {
#ifdef BAMM_PLAIN_LDS
    plain_lds_add(byte_addr, v, pad_mask & nz_mask);
#else
    unsigned long long saved;
    asm volatile("s_and_b64 %0, %3, %4\n\ts_and_saveexec_b64 %0, %0\n\tds_add_u64 %1, %2\n\ts_mov_b64 exec, %0"
                 : "=&s"(saved) : "v"(byte_addr), "v"(v), "s"(pad_mask), "s"(nz_mask) : "memory", "scc");
#endif
}
__device__ __forceinline__ void lds_drain() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// One slot of one column.  The long-sequence classes (M > 16) keep 2*M wave masks, more than the
// SGPR file holds; there the predication is left to the compiler (hipcc's spill code around the
// hand-written exec sequence miscounted at M = 28).
template <int M>
__device__ __forceinline__ void lds_add_slot(uint32_t byte_addr, unsigned long long v, unsigned long long pad_mask,
                                             unsigned long long nz_mask, bool pad_ok) {
    if constexpr (M <= 16) {
        lds_add_u64_masked(byte_addr, v, pad_mask, nz_mask);
    } else {
#if defined(__HIP_DEVICE_COMPILE__)
        typedef __attribute__((address_space(3))) unsigned long long lds_u64;
        if (pad_ok && v != 0ull) atomicAdd((unsigned long long*)(lds_u64*)(size_t)byte_addr, v);
#endif
    }
}

template <int CTRL>
__device__ __forceinline__ float dpp_f(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, false));
}
// sum over the 64 lanes without touching the LDS pipe: rotate-adds inside each row of 16 lanes
// (row_ror:8/4/2/1), then the four row sums through SGPRs.  Same value in every lane.
__device__ __forceinline__ float wave_sum(float x) {
    x += dpp_f<0x128>(x);
    x += dpp_f<0x124>(x);
    x += dpp_f<0x122>(x);
    x += dpp_f<0x121>(x);
    const int xi = __builtin_bit_cast(int, x);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(xi, 0));
    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(xi, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(xi, 32));
    const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(xi, 48));
    return (r0 + r1) + (r2 + r3);
}

// ---- sequence decode: M consecutive positions of one sequence into registers -------------
// y[m] = kmer_[p] mod Y for p = lane*M+m (Sequence.cpp:35-41), PAD (= Y) where p >= limit.
template <int M>
__device__ __forceinline__ void decode_positions(const SeqView& sv, uint32_t seq, uint32_t L,
                                                 uint32_t Y, uint32_t limit, int lane, uint32_t (&y)[M]) {
    constexpr int NSEL = (M + 14) / 16 + 1;  // candidate words per position
    const uint32_t* wp = sv.words + sv.word_off[seq];
    const uint32_t nw = (L + 15u) >> 4;
    const uint32_t p0 = (uint32_t)lane * M;
    const uint32_t wi0 = p0 >> 4;
    uint32_t w[NSEL + 1];  // w[0] = word wi0-1, w[1] = word wi0, ...
    w[0] = (wi0 >= 1u && wi0 - 1u < nw) ? wp[wi0 - 1u] : 0u;
#pragma unroll
    for (int i = 0; i < NSEL; i++) w[i + 1] = (wi0 + i < nw) ? wp[wi0 + i] : 0u;
#pragma unroll
    for (int m = 0; m < M; m++) {
        const uint32_t p = p0 + m;
        const uint32_t sel = (p >> 4) - wi0;
        uint32_t lo = w[1], hi = w[0];
#pragma unroll
        for (int c = 1; c < NSEL; c++) {
            lo = (sel == (uint32_t)c) ? w[c + 1] : lo;
            hi = (sel == (uint32_t)c) ? w[c] : hi;
        }
        const uint32_t sh = 30u - 2u * (p & 15u);
        y[m] = __builtin_amdgcn_alignbit(hi, lo, sh) & (Y - 1u);
    }
    // positions whose k-mer the 2-bit stream cannot express (N randomisation, Sequence.cpp:38)
    const uint64_t e0 = sv.exc_off[seq], e1 = sv.exc_off[seq + 1];
    for (uint64_t e = e0; e < e1; e++) {
        const uint2 x = sv.exc[e];
        if constexpr (M >= 10 && M <= 32) {                  // one indexed register write in the owning lane (see decode_raw)
            const uint32_t pos = __builtin_amdgcn_readfirstlane(x.x), val = __builtin_amdgcn_readfirstlane(x.y);
            const uint32_t owner = pos / (uint32_t)M, slot = pos % (uint32_t)M;
            if ((uint32_t)lane == owner) y[slot] = val;
        } else {
            const int mm = (int)x.x - (int)p0;
#pragma unroll
            for (int m = 0; m < M; m++) y[m] = (mm == m) ? x.y : y[m];
        }
    }
#pragma unroll
    for (int m = 0; m < M; m++) y[m] = (p0 + m < limit) ? y[m] : Y;
}

__device__ __forceinline__ uint32_t pick_sequence(const SeqView& sv, uint32_t t) {
    return sv.idx ? sv.idx[t] : t;
}

// The same decode split in two so that the HBM loads of sequence t+1 are in flight while
// sequence t is being processed (a wave owns one sequence at a time; without this every
// sequence starts with a chain of dependent global loads).
// (the sequence's first kRawExc exceptions travel with it: read one after the other inside decode_raw, each a global round
// trip behind the previous one's use, they were the longest dependent chain of a sequence at k = 4 -- three or four per
// double-stranded sequence at the strand junction)
constexpr int kRawExc = 6;
template <int M>
struct RawSeq {
    static constexpr int NSEL = (M + 14) / 16 + 1;
    uint32_t seq, L;
    uint32_t w[NSEL + 1];
    uint64_t e0, e1;
    uint2 ex[kRawExc];
    bool ok;
};
template <int M>
__device__ __forceinline__ RawSeq<M> fetch_seq(const SeqView& sv, uint32_t t, int lane) {
    RawSeq<M> r;
    r.seq = pick_sequence(sv, t);
    r.ok = !(sv.mask && !sv.mask[r.seq]);
    r.L = sv.len[r.seq];
    const uint32_t* wp = sv.words + sv.word_off[r.seq];
    const uint32_t nw = (r.L + 15u) >> 4;
    const uint32_t wi0 = ((uint32_t)lane * M) >> 4;
    r.w[0] = (wi0 >= 1u && wi0 - 1u < nw) ? wp[wi0 - 1u] : 0u;
#pragma unroll
    for (int i = 0; i < RawSeq<M>::NSEL; i++) r.w[i + 1] = (wi0 + i < nw) ? wp[wi0 + i] : 0u;
    r.e0 = sv.exc_off[r.seq];
    r.e1 = sv.exc_off[r.seq + 1];
#pragma unroll
    for (int i = 0; i < kRawExc; i++) r.ex[i] = (r.e0 + (uint64_t)i < r.e1) ? sv.exc[r.e0 + (uint64_t)i] : make_uint2(0xffffffffu, 0u);
    return r;
}
template <int M>
__device__ __forceinline__ void decode_raw(const RawSeq<M>& r, const SeqView& sv, uint32_t Y, uint32_t limit, int lane,
                                           uint32_t (&y)[M]) {
    constexpr int NSEL = RawSeq<M>::NSEL;
    const uint32_t p0 = (uint32_t)lane * M;
    const uint32_t wi0 = p0 >> 4;
#pragma unroll
    for (int m = 0; m < M; m++) {
        const uint32_t p = p0 + m;
        const uint32_t sel = (p >> 4) - wi0;
        uint32_t lo = r.w[1], hi = r.w[0];
#pragma unroll
        for (int c = 1; c < NSEL; c++) {
            lo = (sel == (uint32_t)c) ? r.w[c + 1] : lo;
            hi = (sel == (uint32_t)c) ? r.w[c] : hi;
        }
        const uint32_t sh = 30u - 2u * (p & 15u);
        y[m] = __builtin_amdgcn_alignbit(hi, lo, sh) & (Y - 1u);
    }
    // N exceptions (Sequence.cpp:38): the ones that came with the sequence, then whatever lies beyond them.  An exception
    // is the same (position, value) in every lane: position and value go to SGPRs, the owning lane is position / M and the
    // slot position mod M is wave-uniform, so the patch is ONE indexed register write under the owner's exec mask
    // (s_set_gpr_idx_on + v_mov) behind a scalar branch -- not a compare and a select per slot (32 VALU instructions and 16
    // wait states per exception, absent ones included: a sixth of the E pass's VALU work at k = 4).
    // (10 to 32 positions per lane: hipcc indexes register arrays of up to 32 dwords, a longer one would move to scratch
    // memory; below 10 the selects are the cheaper of the two -- k = 5, W = 12 at 7 per lane: 1.645 against 1.619 ms)
    auto patch = [&](uint32_t pos_v, uint32_t val_v) {
        if constexpr (M >= 10 && M <= 32) {
            const uint32_t pos = __builtin_amdgcn_readfirstlane(pos_v);
            if (pos != 0xffffffffu) {
                const uint32_t owner = pos / (uint32_t)M, slot = pos % (uint32_t)M;
                const uint32_t val = __builtin_amdgcn_readfirstlane(val_v);
                if ((uint32_t)lane == owner) y[slot] = val;
            }
        } else {
            const uint32_t mm = pos_v - p0;                  // 0xffffffff - p0 never lands in [0, M): lane * M <= 8128
#pragma unroll
            for (int m = 0; m < M; m++) y[m] = (mm == (uint32_t)m) ? val_v : y[m];
        }
    };
#pragma unroll
    for (int i = 0; i < kRawExc; i++) patch(r.ex[i].x, r.ex[i].y);
    for (uint64_t e = r.e0 + (uint64_t)kRawExc; e < r.e1; e++) {
        const uint2 x = sv.exc[e];
        patch(x.x, x.y);
    }
#pragma unroll
    for (int m = 0; m < M; m++) y[m] = (p0 + m < limit) ? y[m] : Y;
}

}  // namespace
}  // namespace bamm
