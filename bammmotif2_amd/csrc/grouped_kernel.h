// gfx950 kernel of the fused EM pass with GROUPED motif columns (orders K <= 2).
//
// Same computation as k_em_seq (kernels.hip): EM::EStep refinement/EM.cpp:149-196, EM::MStep
// EM.cpp:231-243 and the sum over r of EM.cpp:509-513 -- file:line relative to /root/reference/src.
//
// k_em_seq is bound by the LDS pipe: per (position, column) one odds-table gather and one count
// add.  Here G neighbouring columns share ONE table row.  The row index of position p is the
// (K+G)-mer ending at p -- 5 bases (1024 rows, G = 5-K) when the tables fit the 160 KiB of a CU,
// else 4 bases (256 rows, G = 4-K) -- and the table entry of group t is the product of its G column
// odds, so a window costs T = ceil(W/G) gathers / adds instead of W (7 instead of 20 at K=2, W=20):
//
//   U_t(p) = U_{t-1}(p-G) * sG[row(p)][t],      row(p) = kmer_[p] mod 4^(K+G)
//   nG[t][row(p)] += r(window)                  (marginalised to n[j][y] once per block)
//
// What keeps this exact:
//   * EM.cpp:167 truncation (positions >= L-W+1 take no part): a group cut by that edge only carries
//     its leading columns.  Two table layouts (grp_geometry, chosen per launch): "partial" table rows
//     for those G-1 group ends, or per-wave virtual rows like the ones below; beyond the edge the
//     neutral row.
//   * N randomisation (Sequence.cpp:38): next to an exception the k-mers of neighbouring positions
//     disagree, so no (K+G)-mer describes the group.  Those group ends (a handful per sequence:
//     the strand junction) get per-wave VIRTUAL rows: a few "fix" lanes compute their G-column
//     products from the single-column table before the chain starts (the exact y of the positions
//     involved travels inline with the sequence record), and after the M-step move what the
//     virtual count rows collected into single-column bins.  The chain itself never sees an
//     exception.  Sequences whose exceptions span more than the virtual rows go through k_em_seq
//     instead (bamm_em_create splits the buckets).
//   * counts are 64-bit fixed point (2^-40) as in k_em_seq: sums are exact and order-free, so the
//     result is bit-identical to k_em_seq's for the same responsibilities.
// Window products are rounded in a different order than the reference's left-to-right product
// (groups first): relative difference of a few 2^-24, inside the 1e-5 parity bar.
//
// What makes it fast beyond the smaller instruction count: wave priorities by
// phase (the LDS-bound M-step ahead of the VALU-bound E-step), table rows padded to an odd number of
// quads (rows start on all 16 bank-quads), a straight-line E-chain, the group index of the count
// table in the add's immediate offset.
//
// Things measured and left out (DESIGN.md sections 4 and 7): a compacted list of the non-zero windows for the
// M-step (built three ways in round 1, and its LDS traffic alone measured in round 2: a predicated LDS write or
// add costs 6.4 cycles however few lanes take part, so 14 full adds plus the list's writes, row reads and waits
// never beat the 49 sparse ones); the virtual-row counts as no-return atomics on an HBM table instead of the
// LDS one (a CU retires one scattered device-scope atomic per ~18 ns); per-step guards on the run-time group
// count in the E-chain (each merge point costs M register moves: the chain is straight-line over padded
// neutral slots instead).

#pragma once
#include "device_utils.h"
#include "update_kernel.h"

#include <algorithm>
#include <cstdlib>

// the longest length class (positions per lane) whose kernels carry the fused model update in their prologue
#ifndef BAMM_FUSE_MAX_M
#define BAMM_FUSE_MAX_M 16
#endif

namespace bamm {
namespace {

__device__ __forceinline__ void wave_lds_sync() {
    // DS operations of one wave retire in order; this only pins the compiler's ordering
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Entry of the fix lanes' log (kernels with no LDS for the bins of their virtual rows): 8 bytes
//   low word  = the sum as a float.  It is what ONE window added to a virtual count cell: to_fixed40_pre() of an
//               fp32 responsibility, i.e. an integer of at most 24 significant bits -- the conversion to float and
//               back is exact, so the epilogue folds the very integers the kernel without a log would have added
//   high word = which single-column bins [j][y] it belongs to, as the kernel encodes them (group + the y of each of
//               the group's columns; a y field >= Y means "no bin": neutral column, beyond the EM.cpp:167 edge, resident)
// One 8-byte store per logging lane and sequence in the loop, one 8-byte load per entry in the block epilogue
// (12-byte entries with explicit 13-bit bins until round 3: a third more bytes both ways).
typedef unsigned long long GrpLogEntry;
__device__ __forceinline__ void grp_log_store(GrpLogEntry* log, uint32_t idx, unsigned long long sum, uint32_t code) {
    log[idx] = ((unsigned long long)code << 32) | (unsigned long long)__float_as_uint(__ull2float_rn(sum));
}
__device__ __forceinline__ void grp_log_load(const GrpLogEntry* log, uint32_t idx, unsigned long long& sum, uint32_t& code) {
    // written by this wave earlier in the launch: read from L2
    const unsigned long long e = __hip_atomic_load(log + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    sum = (unsigned long long)__uint_as_float((uint32_t)e);
    code = (uint32_t)(e >> 32);
}

template <int I> struct IntC { static constexpr int value = I; };
template <int N, int I = 0, class F>
__device__ __forceinline__ void static_for(F&& f) {           // f(IntC<0>{}), f(IntC<1>{}), ... : compile-time indices
    if constexpr (I < N) { f(IntC<I>{}); static_for<N, I + 1>(f); }
}

// One predicated 64-bit LDS add with the predicate in an SGPR pair: exec <- exec & mask (the incoming
// exec is saved, not assumed to be all lanes), ds_add_u64, exec <- saved.  Same three instructions as
// forcing exec back to -1, but a divergent or partial-wave caller keeps its masked-off lanes off.
template <int OFF>
__device__ __forceinline__ void lds_add_u64_exec(uint32_t byte_addr, unsigned long long v, unsigned long long mask) {
#ifdef BAMM_PLAIN_LDS
    plain_lds_add(byte_addr + (uint32_t)OFF, v, mask);
#else
    unsigned long long saved;
    asm volatile("s_and_saveexec_b64 %0, %3\n\tds_add_u64 %1, %2 offset:%4\n\ts_mov_b64 exec, %0"
                 : "=&s"(saved) : "v"(byte_addr), "v"(v), "s"(mask), "n"(OFF) : "memory", "scc");
#endif
}

template <int OFF>
__device__ __forceinline__ f32x4 lds_read_b128_off(uint32_t byte_addr) {
#ifdef BAMM_PLAIN_LDS
    return lds_load<f32x4>(byte_addr + (uint32_t)OFF);
#else
    f32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(byte_addr), "n"(OFF));
    return v;
#endif
}
template <int OFF>
__device__ __forceinline__ float lds_read_b32_off(uint32_t byte_addr) {
#ifdef BAMM_PLAIN_LDS
    return lds_load<float>(byte_addr + (uint32_t)OFF);
#else
    float v;
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(byte_addr), "n"(OFF));
    return v;
#endif
}

template <int M>
struct RawSeqG {
    static constexpr int NSEL = (M + 14) / 16 + 1;
    uint32_t seq, L;
    uint32_t w[NSEL + 1];
    uint4 xr;
    bool ok;
};

template <int M>
__device__ __forceinline__ RawSeqG<M> fetch_seq_g(const SeqView& sv, const uint4* xrec, uint32_t t, int lane) {
    RawSeqG<M> r;
    r.seq = pick_sequence(sv, t);
    r.ok = !(sv.mask && !sv.mask[r.seq]);
    r.L = sv.len[r.seq];
    r.xr = xrec[r.seq];
    // words past the sequence's end only feed positions >= L, whose rows are overridden; the
    // stream buffer carries 80 words of slack behind the last sequence (bamm_seqs_upload)
    const uint32_t* wp = sv.words + sv.word_off[r.seq];
    const uint32_t wi0 = ((uint32_t)lane * M) >> 4;
    r.w[0] = (wi0 >= 1u) ? wp[wi0 - 1u] : 0u;
#pragma unroll
    for (int i = 0; i < RawSeqG<M>::NSEL; i++) r.w[i + 1] = wp[wi0 + i];
    return r;
}

// The sequence record's y fields (GrpKernelArgs::xrec: one bit string over the words y, z, w, FB bits per field): 32 bits
// of it starting at field k0.  A fix lane's G consecutive fields come out of ONE funnel shift of two neighbouring
// words -- the per-field word selects it replaces hung on eight loop-invariant lane masks that hipcc kept in SGPRs,
// spilled, and fetched back with 16 v_readlane per sequence.
template <int FB>
__device__ __forceinline__ uint32_t xrec_fields(uint32_t wy, uint32_t wz, uint32_t ww, uint32_t k0) {   // by value: no stack copy of the record
    const uint32_t o = (uint32_t)FB * k0;
    const bool first = o < 32u, second = o < 64u;
    const uint32_t lo = first ? wy : (second ? wz : ww);
    const uint32_t hi = first ? wz : (second ? ww : 0u);
    return __builtin_amdgcn_alignbit(hi, lo, o & 31u);
}

// One step of the E-step chain: slots move up by G, the G lowest come from the previous lane.
template <int M, int G>
__device__ __forceinline__ void grp_step(float (&U)[M], const float (&f)[M]) {
    float cy[G];
#pragma unroll
    for (int c = 0; c < G; c++) cy[c] = mul_wave_shr1(f[c], U[M - G + c]);     // before the slots are overwritten
#pragma unroll
    for (int m = M - 1; m >= G; m--) U[m] = U[m - G] * f[m];
#pragma unroll
    for (int m = 0; m < G; m++) U[m] = cy[m];
}

// E-step chain over NQ quads of group slots, straight-line: EVERY slot of a quad is multiplied in
// (the slots in front of the first real group hold 1.0f), so no branch sits between two steps and
// the shift by G slots per step is pure register renaming.  A guard per step (T is a run-time
// value) makes the compiler move all M registers at every merge point.
template <int M, int G, int NQ>
__device__ __forceinline__ void grp_chain(const uint32_t (&ra)[M], float (&U)[M]) {
    static_assert(NQ >= 1 && NQ <= 4, "quads");
    // 56 / 64 positions per lane: 4*M registers in flight per quad do not fit the register file (hipcc then parks
    // destinations it believes ready in AGPRs: round 1's wrong results at these lengths).  One slot at a time
    // instead: M ds_read_b32 in flight, four times the gathers at a third of the cost each.
    if constexpr (M > 48) {
#define BAMM_GRP_SLOT(SL)                                                                       \
        if constexpr (4 * NQ > (SL)) {                                                          \
            float f[M];                                                                         \
            _Pragma("unroll") for (int m = 0; m < M; m++) f[m] = lds_read_b32_off<(SL) * 4>(ra[m]); \
            lds_wait(f);                                                                        \
            if constexpr ((SL) == 0) {                                                          \
                _Pragma("unroll") for (int m = 0; m < M; m++) U[m] = f[m];                      \
            } else {                                                                            \
                grp_step<M, G>(U, f);                                                           \
            }                                                                                   \
        }
        BAMM_GRP_SLOT(0) BAMM_GRP_SLOT(1) BAMM_GRP_SLOT(2) BAMM_GRP_SLOT(3) BAMM_GRP_SLOT(4) BAMM_GRP_SLOT(5)
        BAMM_GRP_SLOT(6) BAMM_GRP_SLOT(7) BAMM_GRP_SLOT(8) BAMM_GRP_SLOT(9) BAMM_GRP_SLOT(10) BAMM_GRP_SLOT(11)
        BAMM_GRP_SLOT(12) BAMM_GRP_SLOT(13) BAMM_GRP_SLOT(14) BAMM_GRP_SLOT(15)
#undef BAMM_GRP_SLOT
        return;
    }
#define BAMM_GRP_QUAD(JQ)                                                                       \
    if constexpr (NQ > (JQ)) {                                                                  \
        f32x4 sv[M];                                                                            \
        _Pragma("unroll") for (int m = 0; m < M; m++) sv[m] = lds_read_b128_off<(JQ) * 16>(ra[m]); \
        lds_wait(sv);                                                                        \
        float f[M];                                                                             \
        if constexpr ((JQ) == 0) {                                                              \
            _Pragma("unroll") for (int m = 0; m < M; m++) U[m] = sv[m].x;                       \
        } else {                                                                                \
            _Pragma("unroll") for (int m = 0; m < M; m++) f[m] = sv[m].x;                       \
            grp_step<M, G>(U, f);                                                               \
        }                                                                                       \
        _Pragma("unroll") for (int m = 0; m < M; m++) f[m] = sv[m].y;                           \
        grp_step<M, G>(U, f);                                                                   \
        _Pragma("unroll") for (int m = 0; m < M; m++) f[m] = sv[m].z;                           \
        grp_step<M, G>(U, f);                                                                   \
        _Pragma("unroll") for (int m = 0; m < M; m++) f[m] = sv[m].w;                           \
        grp_step<M, G>(U, f);                                                                   \
    }
    BAMM_GRP_QUAD(0)
    BAMM_GRP_QUAD(1)
    BAMM_GRP_QUAD(2)
    BAMM_GRP_QUAD(3)
#undef BAMM_GRP_QUAD
}

// WITH_TAIL: the instantiation a launch takes when its pass ends in the in-kernel all-reduce (peer.world > 1).  A variant of
// its own, not a run-time branch of the one kernel: compiled into the single-GPU kernel the tail's mere presence cost the
// k = 1 pass 0.8 % (register allocation of the sequence loop; profiles/r05_ab_peer_check_behind_prologue_and_grp_tail.txt).
template <int M, int G, int KG, bool ACCUM, bool WRITE_R, int THREADS, bool WITH_TAIL = false>
__global__ void __launch_bounds__(THREADS) k_em_grp(GrpKernelArgs ga) {
    static_assert(M >= G, "a group must not span more than two lanes");
    // K = 3 (the only order with two columns per 5-mer row): odds and count tables of 10+ groups leave no LDS
    // for the bins of the virtual rows (41 KB) and often none for the single-column table.  Both serve the few
    // fix lanes only.  The single-column table is read through L2 when the geometry finds no room for it.  What
    // a fix lane takes out of its virtual count row is LOGGED (the non-zero sums of a sequence, compacted: 12
    // bytes each, plain stores) and folded in the block epilogue, when the odds table's LDS is free for the
    // bins.  Adding those counts
    // straight into the pass's accumulator with global atomics (round 2's first version) cost 0.5-1.6 ms per
    // pass: a CU retires one scattered device-scope atomic per ~18 ns (tools/atomic_scatter_bench.hip), and a
    // sequence issued 14-19 of them.  The sequence record carries 10-bit y fields (Y = 256).
    constexpr bool FIXG = (KG - G == 3);
    // the in-kernel all-reduce (update_kernel.h: peer_allreduce_tail) is built into the classes that carry the fused update
    constexpr bool PEER = WITH_TAIL && ACCUM && !FIXG && M <= BAMM_FUSE_MAX_M;
    if (ga.e.stop != nullptr && *ga.e.stop != 0u) return;    // optimize(): the stop rule fired in an earlier pass
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const EmKernelArgs& a = ga.e;
    const GrpGeom& g = ga.g;
    const uint32_t W = a.W, Y = a.Y, Ys = a.Y + 1u;
    const uint32_t T = g.T, Ts = g.Ts, Tq = g.Tq, Rtot = g.Rtot, Rn = g.Rn, delta = g.delta;
    const uint32_t pad = 4u * Tq - T;                     // neutral table slots in front of the first real group
    float* sg = reinterpret_cast<float*>(lds_raw + g.off_sg);                             // [Rtot][Tq][4]
    // [W][Y+1]; K = 3: in LDS when the geometry found room for it (g.cap), else the global table read through L2.
    // Two explicit paths (a pointer chosen at run time would be generic: flat loads count on both memory counters)
    const float* s1_lds = reinterpret_cast<const float*>(lds_raw + g.off_s1);
    const bool s1_global = FIXG && g.cap == 0u;
    auto s1_at = [&](uint32_t idx) -> float { return s1_global ? a.s[idx] : s1_lds[idx]; };
    double* stat_lds = reinterpret_cast<double*>(lds_raw + g.off_stat);                   // [16][3]
    // count table [Rtot][C][Ts] (Ts = T or T | 1: geometry), groups stored last to first: the M-step walks them in that order and
    // reaches a (row, copy)'s next group through the add's immediate offset
    unsigned long long* ng = reinterpret_cast<unsigned long long*>(lds_raw + g.off_ng);
    unsigned long long* n1 = reinterpret_cast<unsigned long long*>(lds_raw + g.off_n1);   // [W][Y]: counts of the virtual rows
    const uint32_t logC = ACCUM ? a.logC : 0u;

    // the wave's first sequence is on its way from HBM while the block builds its tables
    const int lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t waves_per_block = blockDim.x >> 6;
    const uint32_t total_waves = gridDim.x * waves_per_block;
    uint32_t t = blockIdx.x * waves_per_block + wave;
    RawSeqG<M> nxt{};
    if (t < a.sv.count) nxt = fetch_seq_g<M>(a.sv, ga.xrec, t, lane);

    // ---- block prologue: single-column table, grouped table, zeroed counts.  The loads of `s` are issued
    // first and everything that does not need them (zeroed counts, neutral slots) runs under their latency.
    const bool s1_to_lds = !FIXG || g.cap != 0u;
    // fused update (update_kernel.h; planned for K <= 2 and up to 16 positions per lane only): the previous pass's model update runs here, in every
    // block, its scratch where the count table will be; the odds table lands in the single-column table's place and
    // in this block's global copy for the fix lanes.  Inside optimize() a fired stop rule ends the block here.
    [[maybe_unused]] float q_fused = 0.0f;
    bool fused_now = false;
    if constexpr (ACCUM && !FIXG && M <= BAMM_FUSE_MAX_M) {  // up to 1024 positions: beyond that a launch dwarfs the update's 6 us
        if (ga.fused) {
            const UpdateOut uo = model_update_lds<false>(ga.upd, lds_raw + ga.upd_off, reinterpret_cast<float*>(lds_raw + g.off_s1),
                                                         blockIdx.x == 0);
            if (uo.fired) {                                  // block-uniform: the model is final, no pass follows
                if (blockIdx.x == 0) { __syncthreads(); publish_odds(ga.upd, reinterpret_cast<const float*>(lds_raw + g.off_s1), nullptr, true); }
                return;
            }
            q_fused = uo.q;
            fused_now = true;
            __syncthreads();                                 // the scratch becomes the count table below
            publish_odds(ga.upd, reinterpret_cast<const float*>(lds_raw + g.off_s1), ga.s_block + (size_t)blockIdx.x * (W * Ys), blockIdx.x == 0);
        }
    }
    const float* const sfix = fused_now ? ga.s_block + (size_t)blockIdx.x * (W * Ys) : a.s;
    const bool s1_in_regs = !fused_now && s1_to_lds && W * Ys <= 2u * blockDim.x;      // at most two cells per thread: K <= 2 at usual widths
    float s1_r0 = 1.0f, s1_r1 = 1.0f;
    if (s1_in_regs) {
        if (threadIdx.x < W * Ys) s1_r0 = a.s[threadIdx.x];
        if (threadIdx.x + blockDim.x < W * Ys) s1_r1 = a.s[threadIdx.x + blockDim.x];
    }
    if (ACCUM) {
        for (uint32_t i = threadIdx.x; i < (Ts * Rtot) << logC; i += blockDim.x) ng[i] = 0ull;
        if constexpr (!FIXG)
            for (uint32_t i = threadIdx.x; i < W * Y; i += blockDim.x) n1[i] = 0ull;
    }
    // the neutral row and the virtual rows as a whole (the real slots of a virtual row are rewritten per sequence)
    for (uint32_t i = threadIdx.x; i < (Rtot - Rn) * 4u * Tq; i += blockDim.x) {
        const uint32_t row = Rn + i / (4u * Tq), slot = i % (4u * Tq);
        sg[row * g.rowstride + slot] = 1.0f;
    }
    if (s1_to_lds && !fused_now) {
        float* s1w = reinterpret_cast<float*>(lds_raw + g.off_s1);
        if (s1_in_regs) {
            if (threadIdx.x < W * Ys) s1w[threadIdx.x] = s1_r0;
            if (threadIdx.x + blockDim.x < W * Ys) s1w[threadIdx.x + blockDim.x] = s1_r1;
        } else {
            for (uint32_t i = threadIdx.x; i < W * Ys; i += blockDim.x) s1w[i] = a.s[i];
        }
        __syncthreads();
    }
    // a table row per thread (1024 full rows = one each at 1024 threads; the neutral row was filled above), its T
    // groups in a loop: the G column odds of a group multiplied in column order
    for (uint32_t row = threadIdx.x; row < Rn; row += blockDim.x) {
        uint32_t nreal = 0, code = 0;                          // leading positions of the group that are real
        if (row < g.Rf) { nreal = G; code = row; }
#pragma unroll
        for (uint32_t d = 0; d < (uint32_t)(G - 1); d++)       // compile-time indices: no scratch copy of the arrays
            if (d < g.np && row >= g.base[d] && row < g.base[d] + g.psize[d]) { nreal = (uint32_t)G - 1u - d; code = row - g.base[d]; }
        uint32_t yc[G];
#pragma unroll
        for (int c = 0; c < G; c++) yc[c] = (code >> (2u * ((nreal - 1u - c) & 15u))) & (Y - 1u);
        float* out = sg + row * g.rowstride;
        for (uint32_t slot = 0; slot < pad; slot++) out[slot] = 1.0f;       // neutral slots in front
        for (uint32_t t = 0; t < T; t++) {
            float f = 1.0f;
#pragma unroll
            for (int c = 0; c < G; c++) {
                const int col = (int)(G * t + c) - (int)delta;
                if ((uint32_t)c < nreal && col >= 0) f *= s1_at((uint32_t)col * Ys + yc[c]);
            }
            out[pad + t] = f;
        }
    }
    __syncthreads();

    if constexpr (PEER) {                                    // an earlier launch gave up waiting for a peer: nothing is added any more
        // (checked behind the prologue, where the scalar round trip through the kernel-argument segment is hidden: mixed_kernel.h)
        const __attribute__((address_space(4))) GrpKernelArgs* kq =
            (const __attribute__((address_space(4))) GrpKernelArgs*)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(kq));
        if (kq->peer.world > 1u && *kq->peer.err != 0u) return;
    }
    const float q = fused_now ? q_fused : *a.q;
    const float one_minus_q = 1.0f - q;
    const uint32_t lane_b = (uint32_t)lane / T, lane_t = (uint32_t)lane - lane_b * T;     // fix-lane roles
    const uint32_t vbase = g.R0 + wave * g.Bv;
    const uint32_t copy = (uint32_t)lane & ((1u << logC) - 1u);

    double llh_acc = 0.0, sumr_acc = 0.0;
    uint32_t seq_cnt = 0;
    uint32_t last_LW1 = 0;
    float pos_i = 0.0f;
    constexpr bool kCachePos = M <= 16;                      // the longer classes have no registers to spare for it
    [[maybe_unused]] float pos_m[kCachePos ? M : 1];
#pragma unroll
    for (int m = 0; m < (kCachePos ? M : 1); m++) pos_m[m] = 0.0f;
    // K = 3: this wave's log of virtual-row counts (see FIXG above): entries {sum, bins of its G columns}
    [[maybe_unused]] GrpLogEntry* my_log = nullptr;
    [[maybe_unused]] uint32_t nlog = 0;
    if constexpr (FIXG && ACCUM)
        my_log = reinterpret_cast<GrpLogEntry*>(ga.fix_log) + (size_t)(blockIdx.x * waves_per_block + wave) * ga.fix_log_cap;

    for (; t < a.sv.count; t += total_waves) {
        const RawSeqG<M> cur = nxt;
        if (t + total_waves < a.sv.count) {                  // prefetch
            // the sequence set's pointers are read again from the kernel-argument segment (scalar loads) instead of
            // living in SGPRs across the loop body: what hipcc spills of them it fetches back with v_readlane, VALU
            // instructions of a VALU-bound loop (mixed_kernel.h: 0.853 -> 0.842 ms per pass)
            const __attribute__((address_space(4))) GrpKernelArgs* kp =
                (const __attribute__((address_space(4))) GrpKernelArgs*)__builtin_amdgcn_kernarg_segment_ptr();
            asm volatile("" : "+s"(kp));                     // opaque per iteration: the loads are not hoisted out of the loop
            SeqView sv2;
            sv2.words = kp->e.sv.words; sv2.word_off = kp->e.sv.word_off; sv2.len = kp->e.sv.len; sv2.pos_off = nullptr;
            sv2.exc_off = nullptr; sv2.exc = nullptr; sv2.mask = kp->e.sv.mask; sv2.idx = kp->e.sv.idx; sv2.count = a.sv.count;
            nxt = fetch_seq_g<M>(sv2, kp->xrec, t + total_waves, lane);
        }
        const uint32_t seq = cur.seq;
        if (WRITE_R && (seq < a.seq_begin || seq >= a.seq_end)) continue;
        if (!cur.ok) continue;
        // Wave priorities by phase (s_setprio): the M-step is LDS bound and issues few VALU instructions,
        // the E-step is VALU bound.  With the M-step waves ahead in the issue arbitration the LDS pipe
        // stays fed while the E-step waves take the VALU slots left over; staggering the E-step's own
        // phases helps as well.  Measured on the bench workload: 1.26 -> 1.13 ms per pass (any order of
        // priorities beats none; this one is the best of those tried).
        __builtin_amdgcn_s_setprio(0);                       // decode, fix lanes
        const uint32_t L = __builtin_amdgcn_readfirstlane(cur.L);   // one sequence per wave: uniform
        const uint32_t LW1 = L - W + 1u;
        const uint32_t p0 = (uint32_t)lane * M;

        // ---- row index of every position; sE = the 32-bit stream window that ends at position LW1-1.  With one
        // stream window per lane (up to 12 positions per lane) the rows themselves are decoded AFTER the fix lanes
        // have issued their reads of the single-column table (decode_rows below): the round trip runs under them.
        uint32_t row[M];
        uint32_t sE;
        constexpr int NSEL = RawSeqG<M>::NSEL;
        constexpr bool kOneWindow = 2 * (M - 1) + 10 <= 32;                   // rows are at most 10 bits (K+G <= 5)
        const uint32_t wi0 = p0 >> 4;
        const uint32_t pE = LW1 - 1u, lpE = pE / (uint32_t)M;                // lane that holds position LW1-1
        [[maybe_unused]] uint32_t X = 0;
        if constexpr (kOneWindow) {
            // one 32-bit window of the stream ending at the lane's LAST position; position p0+m
            // is that window shifted by a compile-time 2*(M-1-m) bits
            const uint32_t pe = p0 + (uint32_t)(M - 1);
            const uint32_t sel = (pe >> 4) - wi0;
            uint32_t lo = cur.w[1], hi = cur.w[0];
#pragma unroll
            for (int c = 1; c < NSEL; c++) {
                lo = (sel == (uint32_t)c) ? cur.w[c + 1] : lo;
                hi = (sel == (uint32_t)c) ? cur.w[c] : hi;
            }
            X = __builtin_amdgcn_alignbit(hi, lo, 30u - 2u * (pe & 15u));
            // lane lpE's window, shifted so that it ends at pE (at least 32 - 2(M-1) >= 10 bits stay valid)
            sE = (uint32_t)__builtin_amdgcn_readlane((int)X, (int)lpE) >> (2u * ((uint32_t)(M - 1) - (pE - lpE * (uint32_t)M)));
        }
        auto decode_rows = [&]() {
            if constexpr (kOneWindow) {
#pragma unroll
                for (int m = 0; m < M; m++)
                    row[m] = (p0 + m < LW1) ? ((X >> (2 * (M - 1 - m))) & (g.Rf - 1u)) : Rn;   // EM.cpp:167
                if (g.np != 0u) {
                    // groups cut by the LW1 edge read partial rows: G-1 positions of the whole sequence,
                    // patched by their lane
#pragma unroll
                    for (int dd = 0; dd < G - 1; dd++) {
                        const uint32_t pp = LW1 + (uint32_t)dd;
                        if (pp < L) {
                            const uint32_t lp = pp / (uint32_t)M, ms = pp - lp * (uint32_t)M;
                            const uint32_t patch = g.base[dd] + ((X >> (2u * ((uint32_t)(M - 1) - ms) + 2u * (dd + 1))) & (g.psize[dd] - 1u));
                            const bool mine = (uint32_t)lane == lp;
#pragma unroll
                            for (int m = 0; m < M; m++) row[m] = (mine && ms == (uint32_t)m) ? patch : row[m];
                        }
                    }
                }
            } else {
                uint32_t vE = 0;
#pragma unroll
                for (int m = 0; m < M; m++) {
                    const uint32_t p = p0 + m;
                    const uint32_t sel = (p >> 4) - wi0;
                    uint32_t lo = cur.w[1], hi = cur.w[0];
#pragma unroll
                    for (int c = 1; c < NSEL; c++) {
                        lo = (sel == (uint32_t)c) ? cur.w[c + 1] : lo;
                        hi = (sel == (uint32_t)c) ? cur.w[c] : hi;
                    }
                    const uint32_t v = __builtin_amdgcn_alignbit(hi, lo, 30u - 2u * (p & 15u));   // kmer_[p] mod 4^16
                    uint32_t r = (p < LW1) ? (v & (g.Rf - 1u)) : Rn;                              // EM.cpp:167
                    if (g.np != 0u && p >= LW1) {
                        const uint32_t d = p - LW1;
#pragma unroll
                        for (int dd = 0; dd < G - 1; dd++)
                            if (d == (uint32_t)dd && p < L) r = g.base[dd] + ((v >> (2u * (dd + 1))) & (g.psize[dd] - 1u));
                    }
                    row[m] = r;
                    vE = (p == pE) ? v : vE;
                }
                sE = (uint32_t)__builtin_amdgcn_readlane((int)vE, (int)lpE);
            }
        };
        if constexpr (!kOneWindow) decode_rows();            // sE comes out of the per-position windows

        // ---- group ends that no table row describes get per-wave VIRTUAL rows:
        //   * next to an N exception (Sequence.cpp:38): B positions from xlo on; the record carries the
        //     exact y of the positions [xlo-G+1, xlo+B), 7 bits each (Y = none)
        //   * cut by the EM.cpp:167 edge: positions LW1 .. LW1+G-2, whose groups keep only the columns
        //     at positions < LW1 (y from the stream window sE; bamm_em_create keeps sequences with
        //     exceptions next to the edge out of this kernel)
        const uint32_t xw = __builtin_amdgcn_readfirstlane(cur.xr.x);
        const uint32_t B = (xw >> 12) & 0xfu;                // group ends next to exceptions (0: none)
        const uint32_t xlo = xw & 0xfffu;
        // group ends cut by the edge that need a virtual row: none when the table has partial rows for them
        const uint32_t nE = g.np != 0u ? 0u : min((uint32_t)(G - 1), L - LW1);
        // the y of the fix lane's G columns (Y = none) in ONE register, FB bits each: it lives until the virtual count rows
        // are read back after the M-step, and these kernels run at the register limit of their block size
        constexpr uint32_t FB = FIXG ? 10u : 7u, FM = (1u << FB) - 1u;
        uint32_t yfix = 0;
#pragma unroll
        for (int c = 0; c < G; c++) yfix |= Y << (FB * (uint32_t)c);
        const bool fixJ = lane_b < B;                         // fix lanes: (b, t) = virtual row b, group t
        const bool fixE = lane_b >= g.Bj && lane_b < g.Bj + nE;
        const bool fix = fixJ || fixE;
        const bool any_fix = (B | nE) != 0u;                 // wave-uniform
        // the fix lanes' G factors come from the single-column table in GLOBAL memory (through L2; the neutral entry
        // y = Y of a column is 1.0f): the loads are issued here, ahead of the row decode, and land under it -- the
        // vector-memory path is idle otherwise, the LDS pipe is what the pass is short of (the LDS copy of the table
        // only serves the prologue's table build): k = 2 0.980 -> 0.954 ms, k = 1 0.881 -> 0.856 ms
        float fs[G];
#pragma unroll
        for (int c = 0; c < G; c++) fs[c] = 1.0f;
        if (any_fix && fix) {
            const uint32_t xfields = xrec_fields<FIXG ? 10 : 7>(cur.xr.y, cur.xr.z, cur.xr.w, lane_b);     // the fields of lane_b .. lane_b + G - 1
#pragma unroll
            for (int c = 0; c < G; c++) {
                const int col = (int)(G * lane_t + c) - (int)delta;
                uint32_t yc, pos;
                if (fixJ) {
                    const uint32_t k = lane_b + (uint32_t)c;           // entry of position xlo-G+1+k
                    pos = xlo + k - (uint32_t)(G - 1);                 // wraps for positions before the sequence
                    yc = (xfields >> ((FIXG ? 10u : 7u) * (uint32_t)c)) & (FIXG ? 0x3ffu : 0x7fu);
                } else {
                    pos = LW1 + (lane_b - g.Bj) - (uint32_t)(G - 1) + (uint32_t)c;
                    yc = (sE >> (2u * ((LW1 - 1u - pos) & 15u))) & (Y - 1u);   // only used when pos < LW1
                }
                if (col < 0 || pos >= LW1) yc = Y;                     // neutral column / EM.cpp:167 (also pos < 0)
                yfix = (yfix & ~(FM << (FB * (uint32_t)c))) | (yc << (FB * (uint32_t)c));
                const uint32_t idx = __umul24((uint32_t)max(col, 0), Ys) + yc;
                fs[c] = sfix[idx];
            }
        }
        if constexpr (kOneWindow) decode_rows();
        if (any_fix) {
#pragma unroll
            for (int m = 0; m < M; m++) {
                const uint32_t k2 = p0 + m - xlo, k3 = p0 + m - LW1;
                if (k2 < B) row[m] = vbase + k2;
                if (k3 < nE) row[m] = vbase + g.Bj + k3;
            }
            if (fix) {
                float f = 1.0f;
#pragma unroll
                for (int c = 0; c < G; c++) f *= fs[c];                // column order; a neutral entry is 1.0f
                sg[__umul24(vbase + lane_b, g.rowstride) + pad + lane_t] = f;
            }
            wave_lds_sync();
        }
        // ---- E-step: slot p after group t holds the product of groups 0..t of the window whose
        // group t ends at p (EM.cpp:167-176); after the last group that is window p-(W-1)
        __builtin_amdgcn_s_setprio(2);                       // chain
        float U[M];
        {
            uint32_t ra[M];
            const uint32_t sg_base = lds_offset(sg);
#pragma unroll
            for (int m = 0; m < M; m++) ra[m] = sg_base + __umul24(row[m], g.rowstride * 4u);   // v_mad_u32_u24: full rate (v_mul_lo_u32 is not)
            switch (Tq) {
                case 1: grp_chain<M, G, 1>(ra, U); break;
                case 2: grp_chain<M, G, 2>(ra, U); break;
                case 3: grp_chain<M, G, 3>(ra, U); break;
                case 4: grp_chain<M, G, 4>(ra, U); break;
                default: if constexpr (M <= 48) {            // more than 16 groups: plain loop (not planned beyond 48 per lane)
                    grp_chain<M, G, 4>(ra, U);
                    for (uint32_t jq = 4; jq < Tq; jq++) {
                        f32x4 sv[M];
#pragma unroll
                        for (int m = 0; m < M; m++) sv[m] = lds_read_b128(ra[m] + jq * 16u);
                        lds_wait(sv);
                        float f[M];
#pragma unroll
                        for (int m = 0; m < M; m++) f[m] = sv[m].x;
                        grp_step<M, G>(U, f);
#pragma unroll
                        for (int m = 0; m < M; m++) f[m] = sv[m].y;
                        grp_step<M, G>(U, f);
#pragma unroll
                        for (int m = 0; m < M; m++) f[m] = sv[m].z;
                        grp_step<M, G>(U, f);
#pragma unroll
                        for (int m = 0; m < M; m++) f[m] = sv[m].w;
                        grp_step<M, G>(U, f);
                    }
                }
            }
        }
        __builtin_amdgcn_s_setprio(1);                       // normalisation, statistics
        float zpart = 0.0f;
        if constexpr (kCachePos) {
            if (LW1 != last_LW1) {                       // EM.cpp:160; one IEEE division per distinct length
                pos_i = q / (float)LW1;
                last_LW1 = LW1;
                // q/LW1 for the slots that hold a window (it ends at p: p + 1 >= W and p < L), 0 for the others:
                // kept per distinct length, one multiply per slot instead of two compares, a select and a multiply
#pragma unroll
                for (int m = 0; m < M; m++) {
                    const uint32_t p = p0 + m;
                    pos_m[m] = ((p + 1u >= W) && (p < L)) ? pos_i : 0.0f;
                }
            }
#pragma unroll
            for (int m = 0; m < M; m++) {
                U[m] = mul_legacy(U[m], pos_m[m]);   // EM.cpp:180; v_mul_legacy: 0 * anything = 0
                zpart += U[m];
            }
        } else {
            if (LW1 != last_LW1) {
                pos_i = q / (float)LW1;
                last_LW1 = LW1;
            }
#pragma unroll
            for (int m = 0; m < M; m++) {
                const uint32_t p = p0 + m;
                const bool valid = (p + 1u >= W) && (p < L);
                U[m] = valid ? U[m] * pos_i : 0.0f;      // EM.cpp:180
                zpart += U[m];
            }
        }
        const float Z = one_minus_q + wave_sum(zpart);   // EM.cpp:154,181
        // 1/Z: v_rcp_f32 (1 ulp) + one Newton step (error well below an ulp; r = U * invZ stays within
        // 1.5 ulp of the reference's U / Z, EM.cpp:185-187)
        float invZ = __builtin_amdgcn_rcpf(Z);
        invZ = fmaf(fmaf(-Z, invZ, 1.0f), invZ, invZ);
        // the M-step wants r in units of the count accumulator: the power-of-two scale rides on 1/Z (exact)
        const float invZs = ACCUM ? invZ * (a.fix_scale * 256.0f) : invZ;    // 2^8: the fixed-point split below starts there
#pragma unroll
        for (int m = 0; m < M; m++) U[m] = U[m] * invZs; // EM.cpp:185-187
        // EM.cpp:195.  v_log_f32 (log2, 1 ulp of its result) times ln 2: as close to logf(Z) as Z itself is
        // known (Z carries half an ulp of its own); the sum runs in fp64
        llh_acc += stat_round_llh((double)(__builtin_amdgcn_logf(Z) * 0.693147180559945309f));
        sumr_acc += stat_round_sumr((double)one_minus_q * (double)invZ);   // sum_i r[i] = 1 - (1-q)/Z  (EM.cpp:509-513), finished below
        seq_cnt++;

        if (WRITE_R) {                                   // EM::getR layout: r[L-W-i], i = p-W+1
            float* ro = a.r_out + (a.sv.pos_off[seq] - a.r_base);
#pragma unroll
            for (int m = 0; m < M; m++) {
                const uint32_t p = p0 + m;
                if (p < L) ro[L - 1u - p] = U[m];
            }
        }

        if (ACCUM) {
            // ---- M-step (EM.cpp:236-242), grouped: at group t slot p holds r of the window whose
            // group t ends at p; it goes to nG[t][row(p)].  The neutral row is a sink nobody reads.
            __builtin_amdgcn_s_setprio(3);                   // M-step
            unsigned long long F[M];
#pragma unroll
            for (int m = 0; m < M; m++) F[m] = to_fixed40_pre(U[m]);
            const uint32_t ng_base = lds_offset(ng);

            {
                unsigned long long nz[M];
                uint32_t rad[M];                             // byte address of (row, private copy), group T-1
#pragma unroll
                for (int m = 0; m < M; m++) {
                    nz[m] = __ballot(F[m] != 0ull);          // adding an exact 0 is a no-op: those lanes sit out
                    rad[m] = ng_base + __umul24((row[m] << logC) + copy, Ts * 8u);
                }
                // F is a ring: after s steps logical slot m lives in F[(m + G*s) mod M]; the G slots
                // that arrive from the next lane are shifted in place with one DPP pair each, and their
                // non-zero masks with them (lane l takes lane l+1's value: mask >> 1)
                // up to 32 positions per lane a whole turn of the ring (M steps) is unrolled; the longer classes
                // unroll 16 steps (motifs have rarely more groups) and then turn the ring back by 16 steps' worth
                // (48 positions per lane: 1.5e11 -> 1.9e11 positions/s; at 32 the same change cost 17 %)
                constexpr int UB = M <= 32 ? M : 16;
                for (uint32_t sb = 0; sb < T; sb += UB) {
                    static_for<UB>([&](auto uc) {
                        constexpr int u = decltype(uc)::value;
                        if (sb + u < T) {
                            constexpr int off = (G * u) % M;
#pragma unroll
                            for (int m = 0; m < M; m++)
                                lds_add_u64_exec<8 * u>(rad[m], F[(m + off) % M], nz[(m + off) % M]);
#pragma unroll
                            for (int c = 0; c < G; c++) {
                                const int idx = (M - G + c + G * (u + 1)) % M;
                                F[idx] = wave_shl1_u64(F[idx]);
                                nz[idx] >>= 1;
                            }
                        }
                    });
                    if constexpr (UB < M) {
                        unsigned long long Fr[M], nr[M];
#pragma unroll
                        for (int m = 0; m < M; m++) { Fr[m] = F[(m + G * UB) % M]; nr[m] = nz[(m + G * UB) % M]; }
#pragma unroll
                        for (int m = 0; m < M; m++) { F[m] = Fr[m]; nz[m] = nr[m]; }
                    }
#pragma unroll
                    for (int m = 0; m < M; m++) rad[m] += 8u * UB;
                }
            }
            // ---- virtual count rows -> single-column bins (exact: one window per cell)
            if (any_fix) {
                wave_lds_sync();
                unsigned long long acc = 0ull;
                if (fix) {
                    unsigned long long* cell = ng + ((size_t)((vbase + lane_b) << logC)) * Ts + (T - 1u - lane_t);
                    for (uint32_t c = 0; c < (1u << logC); c++) { acc += cell[(size_t)c * Ts]; cell[(size_t)c * Ts] = 0ull; }
                    if constexpr (!FIXG) {
                        if (acc != 0ull) {
#pragma unroll
                            for (int c = 0; c < G; c++) {
                                const int col = (int)(G * lane_t + c) - (int)delta;
                                const uint32_t yc = (yfix >> (FB * (uint32_t)c)) & FM;
                                if (yc != Y) atomicAdd(&n1[(uint32_t)col * Y + yc], acc);
                            }
                        }
                    }
                }
                if constexpr (FIXG) {                            // the non-zero sums, compacted (GrpLogEntry)
                    static_assert(!FIXG || G == 2, "two bins per entry");
                    const unsigned long long nzm = __ballot(acc != 0ull);
                    if (acc != 0ull) {
                        // code: group (4 bits), then the y of its two columns (9 bits each, Y = 256 = no bin)
                        const uint32_t code = lane_t | ((yfix & FM) << 4) | (((yfix >> (FB * (uint32_t)(G - 1))) & FM) << 13);
                        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(nzm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)nzm, 0u));
                        grp_log_store(my_log, nlog + rank, acc, code);
                    }
                    nlog += (uint32_t)__builtin_popcountll(nzm);
                }
                wave_lds_sync();
            }
        }
    }

    // ---- block epilogue: marginalise the grouped counts to n[j][y], statistics
    lds_drain();
    if (lane == 0) {
        stat_lds[wave * 3 + 0] = llh_acc;
        stat_lds[wave * 3 + 1] = (double)seq_cnt - sumr_acc;
        stat_lds[wave * 3 + 2] = (double)seq_cnt;
    }
    __syncthreads();
    if (a.acc == nullptr) return;                        // getR(): responsibilities only
    if constexpr (FIXG && ACCUM) {
        // the logged virtual-row counts -> single-column bins [j][y], which take the odds table's place (every
        // wave is past its last sequence).  A wave reads back only what it wrote itself, after the barrier above
        // (its stores have left the CU; the loads go to L2).
        for (uint32_t i = threadIdx.x; i < W * Y; i += blockDim.x) n1[i] = 0ull;
        __syncthreads();
        // every lane takes entries of the wave's log, NB loads in flight
        constexpr uint32_t NB = 8;
        for (uint32_t e0 = 0; e0 < nlog; e0 += 64u * NB) {
            unsigned long long acc[NB];
            uint32_t code[NB];
#pragma unroll
            for (uint32_t u = 0; u < NB; u++)
                grp_log_load(my_log, min(e0 + u * 64u + (uint32_t)lane, nlog - 1u), acc[u], code[u]);
#pragma unroll
            for (uint32_t u = 0; u < NB; u++) {
                if (e0 + u * 64u + (uint32_t)lane < nlog) {
                    const uint32_t tt = code[u] & 15u;
#pragma unroll
                    for (int c = 0; c < G; c++) {
                        const uint32_t yc = (code[u] >> (4 + 9 * c)) & 511u;
                        const int col = (int)(G * tt + c) - (int)delta;
                        if (yc < Y) atomicAdd(&n1[(uint32_t)col * Y + yc], acc[u]);
                    }
                }
            }
        }
        __syncthreads();
    }
    if (ACCUM) {
        const uint32_t C = 1u << logC;
        for (uint32_t o = threadIdx.x; o < W * Y; o += blockDim.x) {       // o = y*W + j: consecutive global cells
            const uint32_t yy = o / W, j = o - yy * W, i = j * Y + yy;
            const uint32_t t = (j + delta) / G, c = (j + delta) - t * G;
            unsigned long long acc = n1[i];
            const unsigned long long* tab = ng + (T - 1u - t);          // + ((row << logC) + copy) * T
            // full rows whose position c carries yy: c higher digits, G-1-c lower digits are free -- 4^(G-1) rows
            // whatever c is.  One flat loop with a compile-time trip count (the digit split is data, not loop
            // bounds: lanes of a wave differ in c) and independent loads: as nested loops over (h, l) with
            // per-lane bounds the wave ran the union of the lanes' iterations with every load waited for, 3.6 us
            // per cell and half of a launch's fixed cost.
            {
                constexpr uint32_t NR = 1u << (2u * (uint32_t)(G - 1));
                const uint32_t lowd = 2u * ((uint32_t)G - 1u - c), lmask = (1u << lowd) - 1u;
                unsigned long long part[4] = {0ull, 0ull, 0ull, 0ull};
                // (every lane starts its rows elsewhere: at equal steps the lanes' rows differ in digits that do not reach
                // the bank; integer sums, any order gives the same total)
                const uint32_t rot = ((uint32_t)lane + ((uint32_t)lane >> 2)) & (NR < 16u ? NR - 1u : 15u);
#pragma unroll
                for (uint32_t r0 = 0; r0 < NR; r0++) {
                    const uint32_t r = (r0 & ~15u) | ((r0 + rot) & (NR < 16u ? NR - 1u : 15u));
                    const uint32_t h = r >> lowd, l = r & lmask;
                    // the (K+G)-mer has exactly c digits above y_c, so the mask is a no-op
                    const uint32_t row = ((((h << (2u * (a.K + 1u))) | yy) << lowd) | l) & (g.Rf - 1u);
                    for (uint32_t cc = 0; cc < C; cc++) part[r0 & 3u] += tab[(((size_t)row << logC) + cc) * Ts];
                }
                acc += (part[0] + part[1]) + (part[2] + part[3]);
            }
#pragma unroll
            for (uint32_t d = 0; d < (uint32_t)(G - 1); d++) {   // partial rows: positions c < G-1-d are real
                const uint32_t nreal = (uint32_t)G - 1u - d;
                if (d < g.np && c < nreal) {
                    const uint32_t NRp = 1u << (2u * (nreal - 1u));
                    const uint32_t lowd = 2u * (nreal - 1u - c), lmask = (1u << lowd) - 1u;
                    for (uint32_t r = 0; r < NRp; r++) {
                        const uint32_t h = r >> lowd, l = r & lmask;
                        const uint32_t row = g.base[d] + ((((h << (2u * (a.K + 1u))) | yy) << lowd) | l);
                        for (uint32_t cc = 0; cc < C; cc++) acc += tab[(((size_t)row << logC) + cc) * Ts];
                    }
                }
            }
            if (acc) acc_add(a.acc + o, (long long)acc);
        }
    }
    if (threadIdx.x < 3) {
        double acc = 0.0;
        for (uint32_t w = 0; w < waves_per_block; w++) acc += stat_lds[w * 3 + threadIdx.x];
        acc_add_stat(a.acc, W * Y, threadIdx.x, acc);
    }
    if constexpr (PEER) {
        // in-kernel all-reduce: the last block to get here sums the GPUs' totals.  Its arguments are read from the
        // kernel-argument segment HERE, through a pointer the compiler cannot see through (as k_em_mix does: as ordinary
        // arguments they lived in SGPRs across the sequence loop and cost it 0.8 %, profiles/r04_ab_peer_tail_as_arguments.txt)
        const __attribute__((address_space(4))) GrpKernelArgs* kq =
            (const __attribute__((address_space(4))) GrpKernelArgs*)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(kq));
        if (kq->peer.world > 1u) {
            __syncthreads();                                 // the LDS is nobody's any more
            peer_allreduce_tail(&kq->peer, kq->e.acc, reinterpret_cast<uint32_t*>(lds_raw));
        }
    }
}

template <int M, int G, int KG, int THREADS>
int launch_variant(bool accum, bool write_r, const GrpKernelArgs& a, uint32_t blocks, uint32_t threads, hipStream_t st) {
    if constexpr (M < G) {                                   // grp_geometry never plans a group wider than a lane's run
        (void)accum; (void)write_r; (void)a; (void)blocks; (void)threads; (void)st;
        set_error("grouped kernel: group of %d columns wider than a lane's %d positions", G, M);
        return BAMM_ERR_UNSUPPORTED;
    } else {
        const size_t lds = a.g.lds_bytes;
        int rc;
        if (write_r) {
            if ((rc = allow_lds(reinterpret_cast<const void*>(&k_em_grp<M, G, KG, false, true, THREADS>), lds))) return rc;
            if (blocks == kPrimeOnly) return prime_kernel(reinterpret_cast<const void*>(&k_em_grp<M, G, KG, false, true, THREADS>));
            hipLaunchKernelGGL((k_em_grp<M, G, KG, false, true, THREADS>), dim3(blocks), dim3(threads), lds, st, a);
        } else if (accum) {
            if constexpr (KG - G != 3 && M <= BAMM_FUSE_MAX_M) {             // the classes built with the all-reduce tail
                if (a.peer.world > 1u) {
                    if ((rc = allow_lds(reinterpret_cast<const void*>(&k_em_grp<M, G, KG, true, false, THREADS, true>), lds))) return rc;
                    hipLaunchKernelGGL((k_em_grp<M, G, KG, true, false, THREADS, true>), dim3(blocks), dim3(threads), lds, st, a);
                    return BAMM_OK;
                }
            }
            if (a.peer.world > 1u) { set_error("grouped kernel: this class is not built with the in-kernel all-reduce"); return BAMM_ERR_UNSUPPORTED; }
            if ((rc = allow_lds(reinterpret_cast<const void*>(&k_em_grp<M, G, KG, true, false, THREADS>), lds))) return rc;
            if (blocks == kPrimeOnly) return prime_kernel(reinterpret_cast<const void*>(&k_em_grp<M, G, KG, true, false, THREADS>));
            hipLaunchKernelGGL((k_em_grp<M, G, KG, true, false, THREADS>), dim3(blocks), dim3(threads), lds, st, a);
        } else {
            if ((rc = allow_lds(reinterpret_cast<const void*>(&k_em_grp<M, G, KG, false, false, THREADS>), lds))) return rc;
            if (blocks == kPrimeOnly) return prime_kernel(reinterpret_cast<const void*>(&k_em_grp<M, G, KG, false, false, THREADS>));
            hipLaunchKernelGGL((k_em_grp<M, G, KG, false, false, THREADS>), dim3(blocks), dim3(threads), lds, st, a);
        }
        return BAMM_OK;
    }
}

}  // namespace
}  // namespace bamm

// one length class of the dispatch switch (key = class * 64 + G * 8 + K + G), every (G, K+G) the planner can ask for
#define BAMM_GRP_CASES(idx, M, T)                                                                           \
    case idx * 64 + 2 * 8 + 4: if (int rc = launch_variant<M, 2, 4, T>(accum, write_r, a, blocks, threads, st)) return rc; break;   \
    case idx * 64 + 2 * 8 + 5: if (int rc = launch_variant<M, 2, 5, T>(accum, write_r, a, blocks, threads, st)) return rc; break;   \
    case idx * 64 + 3 * 8 + 4: if (int rc = launch_variant<M, 3, 4, T>(accum, write_r, a, blocks, threads, st)) return rc; break;   \
    case idx * 64 + 4 * 8 + 4: if (int rc = launch_variant<M, 4, 4, T>(accum, write_r, a, blocks, threads, st)) return rc; break;   \
    case idx * 64 + 3 * 8 + 5: if (int rc = launch_variant<M, 3, 5, T>(accum, write_r, a, blocks, threads, st)) return rc; break;   \
    case idx * 64 + 4 * 8 + 5: if (int rc = launch_variant<M, 4, 5, T>(accum, write_r, a, blocks, threads, st)) return rc; break;
