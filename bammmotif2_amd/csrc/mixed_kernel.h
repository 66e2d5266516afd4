// gfx950 kernel of the fused EM pass at K = 2 with MIXED table rows: the motif's first B groups of 3 columns
// on 5-mer rows as in k_em_grp (grouped_kernel.h), its last A = W mod 3 groups of FOUR columns on 6-mer rows.
//
// Same computation as k_em_grp / k_em_seq: EM::EStep refinement/EM.cpp:149-196, EM::MStep EM.cpp:231-243, the
// sum over r of EM.cpp:509-513 -- file:line relative to /root/reference/src.
//
// Why: k_em_grp is bound by LDS instructions, T = ceil(W/3) gathers of group odds and T count adds per
// position, and a predicated ds_add_u64 costs 6.4-6.7 LDS cycles however few lanes take part (DESIGN.md
// section 4).  When W is not a multiple of 3 the last group of the uniform layout is mostly padding: W = 20 is
// 7 groups.  A 6-mer row holds four columns, so 20 = 3+3+3+3 + 4+4 is SIX groups: 42 adds instead of 49 per
// sequence of 7 positions per lane, and one b128 + one b64 gather per position instead of two b128.  An LDS-only
// loop of that mix (tools/lds_mix_bench.hip) runs in 165 ns per sequence and CU against 197 ns.  The price is
// LDS: a wide group takes 16 KB of odds and 32 KB of counts (4096 rows) against 4 + 8 KB, so at most two of
// them fit, the count tables are group-major (no padding cell), and the bins of the virtual rows only exist
// in the block epilogue (the fix lanes log their sums, as k_em_grp does at K = 3).
//
//   U_t(p) = U_{t-1}(p - G_t) * tab_t[row_t(p)]     narrow t < B: G_t = 3, row = kmer_[p] mod 4^5; wide: 4, 4^6
//   cnt_t[row_t(p)] += r(window whose group t ends at p)          (2^-40 fixed point, marginalised per block)
//
// Everything else is k_em_grp's: one wavefront per sequence, M positions per lane, per-wave virtual rows for the
// group ends next to an N exception (Sequence.cpp:38) and for those cut by the EM.cpp:167 edge -- one virtual
// row index serves both tables --, wave priorities by phase, the straight-line chain.
#pragma once
#include "grouped_kernel.h"

namespace bamm {
namespace {

constexpr uint32_t kMixBj = 6u, kMixNe = 3u, kMixBv = kMixBj + kMixNe;   // virtual rows per wave: exceptions, edge

template <int OFF>
__device__ __forceinline__ void lds_add_u64_exec_big(uint32_t byte_addr, unsigned long long v, unsigned long long mask) {
    static_assert(OFF >= 0 && OFF < 65536, "ds offset field");
    lds_add_u64_exec<OFF>(byte_addr, v, mask);
}

template <int M, int A, int NQ, bool ACCUM, bool WRITE_R, int THREADS>
__global__ void __launch_bounds__(THREADS) k_em_mix(GrpKernelArgs ga) {
    static_assert(M >= 4 && 2 * (M - 1) + 12 <= 32, "a wide group within one lane's run, rows out of one stream window");
    static_assert(A == 1 || A == 2, "wide groups");
    constexpr uint32_t WAVES = THREADS / 64, V = WAVES * kMixBv;
    constexpr uint32_t R5N = 1024u, R5V0 = R5N + 1u, R5T = R5V0 + V;          // neutral row, first virtual row, rows
    constexpr uint32_t R6N = 4096u, R6V0 = R6N + 1u, R6T = R6V0 + V;
    constexpr uint32_t Y = 64u, Ys = 65u;                      // K = 2
    if (ga.e.stop != nullptr && *ga.e.stop != 0u) return;    // optimize(): the stop rule fired in an earlier pass
    extern __shared__ __align__(16) unsigned char lds_raw[];
    BAMM_PHASE(0);
    const EmKernelArgs& a = ga.e;
    const GrpGeom& g = ga.g;
    const uint32_t W = a.W, T = g.T, B = g.mixB;             // T = B + A groups
    const uint32_t pad = 4u * NQ - B;                        // neutral slots in front of the first narrow group
    const uint32_t rs5 = g.rowstride;                        // floats per narrow odds row
    float* sg5 = reinterpret_cast<float*>(lds_raw + g.off_sg);                            // [R5T][rs5]
    float* sg6 = reinterpret_cast<float*>(lds_raw + g.off_sg6);                           // [R6T][A]
    // The single-column table [W][Y+1] serves the prologue (staged where the counts will be) and the fix lanes, who
    // read it from global memory through L2: four loads per sequence on the otherwise idle vector-memory path instead
    // of four LDS reads (0.875 -> 0.855 ms), and 5 KB of LDS for the bins below.
    const float* s1_stage = reinterpret_cast<const float*>(lds_raw + g.off_s1);            // prologue only: over the counts
    double* stat_lds = reinterpret_cast<double*>(lds_raw + g.off_stat);                   // [16][3], epilogue only: over sg5
    // count tables, group-major, groups stored LAST TO FIRST (the M-step walks them in that order and reaches the
    // next group through the add's immediate offset): cnt5[B-1-t][row], cnt6[T-1-t][row]
    unsigned long long* cnt5 = reinterpret_cast<unsigned long long*>(lds_raw + g.off_ng);
    unsigned long long* cnt6 = reinterpret_cast<unsigned long long*>(lds_raw + g.off_ng6);
    unsigned long long* n1 = reinterpret_cast<unsigned long long*>(lds_raw + g.off_n1);   // epilogue only: over sg6
    // bins [j][y] of the motif's first n1c columns, resident: what the fix lanes take out of their virtual count rows
    // for those columns is added here; only the other columns' sums go through the log (each logged entry costs a
    // store request in the loop and a read in the epilogue: all of them 0.875 ms, a third of them 0.814 ms)
    unsigned long long* n1p = reinterpret_cast<unsigned long long*>(lds_raw + g.off_wave);
    const uint32_t n1c = ACCUM ? g.wave_bytes / (Y * 8u) : 0u;

    const int lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t total_waves = gridDim.x * WAVES;
    uint32_t t = blockIdx.x * WAVES + wave;
    RawSeqG<M> nxt{};
    if (t < a.sv.count) nxt = fetch_seq_g<M>(a.sv, ga.xrec, t, lane);

    // ---- block prologue
    // fused update (update_kernel.h): the previous pass's model update runs here, in every block, from the all-reduced
    // accumulator -- its odds table lands where the prologue stages it anyway, and in this block's global copy for
    // the fix lanes; block 0 publishes the model.  Inside optimize() a fired stop rule ends the block here.
    [[maybe_unused]] float q_fused = 0.0f;
    {
        float* s1w = reinterpret_cast<float*>(lds_raw + g.off_s1);
        const float* s1 = s1_stage;
        bool staged = false;
        if constexpr (ACCUM) {
            if (ga.fused) {
                const UpdateOut uo = model_update_lds<false>(ga.upd, lds_raw + ga.upd_off, s1w, blockIdx.x == 0);
                if (uo.fired) {                              // block-uniform: the model is final, no pass follows
                    if (blockIdx.x == 0) { __syncthreads(); publish_odds(ga.upd, s1w, nullptr, true); }
                    return;
                }
                q_fused = uo.q;
                staged = true;
            }
        }
        BAMM_PHASE(1);                                       // the fused update is done
        if (!staged)
            for (uint32_t i = threadIdx.x; i < W * Ys; i += THREADS) s1w[i] = a.s[i];
        for (uint32_t i = threadIdx.x; i < (R5T - R5N) * rs5; i += THREADS) sg5[R5N * rs5 + i] = 1.0f;
        for (uint32_t i = threadIdx.x; i < (R6T - R6N) * A; i += THREADS) sg6[R6N * A + i] = 1.0f;
        __syncthreads();
        if constexpr (ACCUM)
            if (staged) publish_odds(ga.upd, s1w, ga.s_block + (size_t)blockIdx.x * (W * Ys), blockIdx.x == 0);
        for (uint32_t row = threadIdx.x; row < R5N; row += THREADS) {      // a group's 3 column odds in column order
            float* out = sg5 + row * rs5;
            for (uint32_t slot = 0; slot < pad; slot++) out[slot] = 1.0f;
            for (uint32_t slot = pad + B; slot < rs5; slot++) out[slot] = 1.0f;
            for (uint32_t tt = 0; tt < B; tt++) {
                float f = 1.0f;
#pragma unroll
                for (int c = 0; c < 3; c++) f *= s1[(3u * tt + c) * Ys + ((row >> (2 * (2 - c))) & 63u)];
                out[pad + tt] = f;
            }
        }
        for (uint32_t row = threadIdx.x; row < R6N; row += THREADS) {
#pragma unroll
            for (int w = 0; w < A; w++) {
                float f = 1.0f;
#pragma unroll
                for (int c = 0; c < 4; c++) f *= s1[(3u * B + 4u * w + c) * Ys + ((row >> (2 * (3 - c))) & 63u)];
                sg6[row * A + w] = f;
            }
        }
        __syncthreads();
        if (ACCUM) {                                         // the staging area becomes the count tables
            for (uint32_t i = threadIdx.x; i < B * R5T; i += THREADS) cnt5[i] = 0ull;
            for (uint32_t i = threadIdx.x; i < A * R6T; i += THREADS) cnt6[i] = 0ull;
            for (uint32_t i = threadIdx.x; i < n1c * Y; i += THREADS) n1p[i] = 0ull;
            __syncthreads();
        }
    }

    BAMM_PHASE(2);                                           // tables built, counts zeroed
    if constexpr (ACCUM) {
        // in-kernel all-reduce: an earlier launch gave up waiting for a peer -- nothing is added any more (the prologue above
        // only read; block 0's publication of the model it computed from sums that never completed is as invalid as the handle,
        // whose next read fails with BAMM_ERR_COMM).  Checked HERE, not at the entry: the scalar round trip through the
        // kernel-argument segment is then behind the table build instead of in front of the first sequence fetch.
        const __attribute__((address_space(4))) GrpKernelArgs* kq =
            (const __attribute__((address_space(4))) GrpKernelArgs*)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(kq));
        if (kq->peer.world > 1u && *kq->peer.err != 0u) return;
    }
    const float q = (ACCUM && ga.fused) ? q_fused : *a.q;
    const float one_minus_q = 1.0f - q;
    const uint32_t lane_b = (uint32_t)lane / T, lane_t = (uint32_t)lane - lane_b * T;     // fix-lane roles: (row, group)
    const bool lane_wide = lane_t >= B;
    const uint32_t lane_G = lane_wide ? 4u : 3u;
    const uint32_t lane_col0 = lane_wide ? 3u * B + 4u * (lane_t - B) : 3u * lane_t;
    const uint32_t vbase5 = R5V0 + wave * kMixBv, vbase6 = R6V0 + wave * kMixBv;
    const uint32_t sg5_base = lds_offset(sg5), sg6_base = lds_offset(sg6);

    // per lane, fixed for the launch: bit 6 (the value Y) of every 7-bit field of `yfix` whose column has a resident bin -- OR-ed
    // into the sequence's codes it marks those columns "nothing to log" in one instruction
    uint32_t resident_fill = 0u;
#pragma unroll
    for (int c = 0; c < 4; c++)
        if (lane_col0 + (uint32_t)c < n1c) resident_fill |= Y << (7 * c);
    double llh_acc = 0.0, sumr_acc = 0.0;
    uint32_t seq_cnt = 0, last_LW1 = 0;
    float pos_i = 0.0f;
    float pos_m[M];
    constexpr bool kCacheSpec = M <= 7 || THREADS < 1024;
    [[maybe_unused]] int spec[kCacheSpec ? M : 1];           // per slot: offset from the neutral row, -1 = a stream row
    [[maybe_unused]] uint32_t spec_LW1 = 0, spec_xw = 0xffffffffu;
#pragma unroll
    for (int m = 0; m < M; m++) pos_m[m] = 0.0f;
#pragma unroll
    for (int m = 0; m < (kCacheSpec ? M : 1); m++) spec[m] = -1;
    [[maybe_unused]] GrpLogEntry* my_log = nullptr;          // the fix lanes' non-zero sums (8-byte entries)
    [[maybe_unused]] uint32_t nlog = 0;
    if constexpr (ACCUM)
        my_log = reinterpret_cast<GrpLogEntry*>(ga.fix_log) + (size_t)(blockIdx.x * WAVES + wave) * ga.fix_log_cap;

    for (; t < a.sv.count; t += total_waves) {
        const RawSeqG<M> cur = nxt;
        // Pointers that are needed once per sequence are read again from the kernel-argument segment (scalar loads,
        // scalar cache) instead of living in SGPRs across the whole loop body: the body is 100 SGPRs over budget, and
        // what hipcc spills it parks in VGPR lanes and fetches back with v_readlane -- VALU instructions of a
        // VALU-bound loop (0.853 -> 0.842 ms per pass at 1M for the sequence set's pointers alone).
        const __attribute__((address_space(4))) GrpKernelArgs* kp =
            (const __attribute__((address_space(4))) GrpKernelArgs*)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(kp));                         // opaque per iteration: the loads are not hoisted out of the loop
        if (t + total_waves < a.sv.count) {
            SeqView sv2;
            sv2.words = kp->e.sv.words; sv2.word_off = kp->e.sv.word_off; sv2.len = kp->e.sv.len; sv2.pos_off = nullptr;
            sv2.exc_off = nullptr; sv2.exc = nullptr; sv2.mask = kp->e.sv.mask; sv2.idx = kp->e.sv.idx; sv2.count = a.sv.count;
            nxt = fetch_seq_g<M>(sv2, kp->xrec, t + total_waves, lane);
        }
        const uint32_t seq = cur.seq;
        if (WRITE_R && (seq < a.seq_begin || seq >= a.seq_end)) continue;
        if (!cur.ok) continue;
        __builtin_amdgcn_s_setprio(0);                       // decode, fix lanes
        const uint32_t L = __builtin_amdgcn_readfirstlane(cur.L);
        const uint32_t LW1 = L - W + 1u;
        const uint32_t p0 = (uint32_t)lane * M;

        // ---- one 32-bit stream window per lane (it ends at the lane's last position); sE = the window ending at LW1-1
        uint32_t X, sE;
        {
            constexpr int NSEL = RawSeqG<M>::NSEL;
            const uint32_t wi0 = p0 >> 4;
            const uint32_t pE = LW1 - 1u, lpE = pE / (uint32_t)M;
            const uint32_t pe = p0 + (uint32_t)(M - 1);
            const uint32_t sel = (pe >> 4) - wi0;
            uint32_t lo = cur.w[1], hi = cur.w[0];
#pragma unroll
            for (int c = 1; c < NSEL; c++) {
                lo = (sel == (uint32_t)c) ? cur.w[c + 1] : lo;
                hi = (sel == (uint32_t)c) ? cur.w[c] : hi;
            }
            X = __builtin_amdgcn_alignbit(hi, lo, 30u - 2u * (pe & 15u));
            sE = (uint32_t)__builtin_amdgcn_readlane((int)X, (int)lpE) >> (2u * ((uint32_t)(M - 1) - (pE - lpE * (uint32_t)M)));
        }

        // ---- virtual rows (one index for both tables): B group ends from xlo on next to an exception, the
        // positions LW1 .. LW1+2 whose groups are cut by the edge.  The fix lanes' reads of the single-column table
        // are issued first and waited for after the rows of all positions are decoded: the wave is not yet bound
        // by the LDS pipe here, the round trip is latency it would otherwise sit out.
        const uint32_t xw = __builtin_amdgcn_readfirstlane(cur.xr.x);
        const uint32_t Bx = (xw >> 12) & 0xfu;
        const uint32_t xlo = xw & 0xfffu;
        const uint32_t nE = min(kMixNe, L - LW1);
        // the y of the fix lane's up to four columns, 7 bits each (Y = none), in ONE register: it lives until the virtual
        // count rows are read back after the M-step, and the kernel runs at the 128-VGPR limit of a 1024-thread block
        uint32_t yfix = Y | (Y << 7) | (Y << 14) | (Y << 21);
        const bool fixJ = lane_b < Bx;
        const bool fixE = lane_b >= kMixBj && lane_b < kMixBj + nE;
        const bool fix = fixJ || fixE;
        float fs[4] = {1.0f, 1.0f, 1.0f, 1.0f};
        const float* const sfix_now = (ACCUM && kp->fused) ? kp->s_block + (size_t)blockIdx.x * (W * Ys) : kp->e.s;
        if (fix) {
            const uint32_t pv = fixJ ? xlo + lane_b : LW1 + (lane_b - kMixBj);         // the row's position
            const uint32_t xfields = xrec_fields<7>(cur.xr.y, cur.xr.z, cur.xr.w, lane_b + 4u - lane_G);     // the fields of the group's columns
#pragma unroll
            for (int c = 0; c < 4; c++) {
                uint32_t yc = Y;                                                       // the column's neutral entry
                const uint32_t colc = min(lane_col0 + (uint32_t)c, W - 1u);
                if ((uint32_t)c < lane_G) {
                    const uint32_t pos = pv - (lane_G - 1u) + (uint32_t)c;           // wraps for positions before the sequence
                    if (fixJ) {                                                        // record fields start at position xlo-3
                        yc = (xfields >> (7u * (uint32_t)c)) & 0x7fu;
                    } else {
                        yc = (sE >> (2u * ((LW1 - 1u - pos) & 15u))) & (Y - 1u);
                    }
                    if (pos >= LW1) yc = Y;                                            // EM.cpp:167 (also pos < 0)
                }
                yfix = (yfix & ~(0x7fu << (7 * c))) | (yc << (7 * c));
                fs[c] = sfix_now[__umul24(colc, Ys) + yc];                             // global, through L1 / L2
            }
        }
        // Which slots do not take their row from the stream -- beyond the EM.cpp:167 edge (neutral row), a
        // virtual row -- depends on (L, the record's first word) only, and sets of one length with the strand
        // junction in one place repeat both: the slot's offset from the neutral row (the virtual rows follow it
        // in both tables; -1: a stream row) is kept until either changes.  2 bit-field extracts, 1 compare, 2 adds
        // and 2 selects per slot instead of 17 instructions.
        // (8 positions per lane at 1024 threads have no registers to spare for it: the offsets are recomputed)
        auto slot_offsets = [&](int (&o)[M]) {
#pragma unroll
            for (int m = 0; m < M; m++) {
                const uint32_t k2 = p0 + m - xlo, k3 = p0 + m - LW1;
                o[m] = (p0 + m < LW1) ? -1 : 0;                                          // EM.cpp:167
                if (k2 < Bx) o[m] = (int)(1u + wave * kMixBv + k2);
                if (k3 < nE) o[m] = (int)(1u + wave * kMixBv + kMixBj + k3);
            }
        };
        uint32_t row5[M], row6[M];
        if constexpr (kCacheSpec) {
            if (LW1 != spec_LW1 || xw != spec_xw) {
                spec_LW1 = LW1; spec_xw = xw;
                slot_offsets(spec);
            }
#pragma unroll
            for (int m = 0; m < M; m++) {
                const bool stream = spec[m] < 0;
                row5[m] = stream ? ((X >> (2 * (M - 1 - m))) & 1023u) : R5N + (uint32_t)spec[m];
                row6[m] = stream ? ((X >> (2 * (M - 1 - m))) & 4095u) : R6N + (uint32_t)spec[m];
            }
        } else {
            int o[M];
            slot_offsets(o);
#pragma unroll
            for (int m = 0; m < M; m++) {
                const bool stream = o[m] < 0;
                row5[m] = stream ? ((X >> (2 * (M - 1 - m))) & 1023u) : R5N + (uint32_t)o[m];
                row6[m] = stream ? ((X >> (2 * (M - 1 - m))) & 4095u) : R6N + (uint32_t)o[m];
            }
        }
        if (fix) {
            const float f = ((fs[0] * fs[1]) * fs[2]) * fs[3];                         // column order; a neutral entry is 1.0f
            if (lane_wide) sg6[(vbase6 + lane_b) * A + (lane_t - B)] = f;
            else sg5[__umul24(vbase5 + lane_b, rs5) + pad + lane_t] = f;
        }
        wave_lds_sync();

        // ---- E-step: narrow groups (straight-line over NQ quads, neutral slots in front), then the wide ones
        __builtin_amdgcn_s_setprio(2);
        float U[M];
        {
            uint32_t ra[M];
#pragma unroll
            for (int m = 0; m < M; m++) ra[m] = sg5_base + __umul24(row5[m], rs5 * 4u);        // v_mad_u32_u24: full rate
            // the wide groups' odds are on their way while the narrow groups are chained (one round trip, not two)
            float2 pr[M];
#pragma unroll
            for (int m = 0; m < M; m++) {
#ifdef BAMM_PLAIN_LDS
                if constexpr (A == 2) pr[m] = lds_load<float2>(sg6_base + row6[m] * 8u);
                else pr[m].x = lds_load<float>(sg6_base + row6[m] * 4u);
#else
                if constexpr (A == 2) asm volatile("ds_read_b64 %0, %1" : "=v"(pr[m]) : "v"(sg6_base + row6[m] * 8u));
                else asm volatile("ds_read_b32 %0, %1" : "=v"(pr[m].x) : "v"(sg6_base + row6[m] * 4u));
#endif
            }
            grp_chain<M, 3, NQ>(ra, U);                      // waits with lgkmcnt(0): the reads above have landed as well
            lds_wait(pr);
            float w0[M];
#pragma unroll
            for (int m = 0; m < M; m++) w0[m] = pr[m].x;
            grp_step<M, 4>(U, w0);
            if constexpr (A == 2) {
                float w1[M];
#pragma unroll
                for (int m = 0; m < M; m++) w1[m] = pr[m].y;
                grp_step<M, 4>(U, w1);
            }
        }
        __builtin_amdgcn_s_setprio(1);                       // normalisation, statistics
        if (LW1 != last_LW1) {                           // EM.cpp:160; one IEEE division per distinct length
            pos_i = q / (float)LW1;
            last_LW1 = LW1;
            // q/LW1 for the slots that hold a window (it ends at p: p + 1 >= W and p < L), 0 for the others: kept
            // per distinct length, one multiply per slot instead of two compares, a select and a multiply
#pragma unroll
            for (int m = 0; m < M; m++) {
                const uint32_t p = p0 + m;
                pos_m[m] = ((p + 1u >= W) && (p < L)) ? pos_i : 0.0f;
            }
        }
        float zpart = 0.0f;
#pragma unroll
        for (int m = 0; m < M; m++) {
            U[m] = mul_legacy(U[m], pos_m[m]);   // EM.cpp:180; v_mul_legacy: 0 * anything = 0
            zpart += U[m];
        }
        const float Z = one_minus_q + wave_sum(zpart);       // EM.cpp:154,181
        float invZ = __builtin_amdgcn_rcpf(Z);
        invZ = fmaf(fmaf(-Z, invZ, 1.0f), invZ, invZ);
        const float invZs = ACCUM ? invZ * (a.fix_scale * 256.0f) : invZ;    // 2^8: the fixed-point split below starts there
#pragma unroll
        for (int m = 0; m < M; m++) U[m] = U[m] * invZs;     // EM.cpp:185-187
        llh_acc += stat_round_llh((double)(__builtin_amdgcn_logf(Z) * 0.693147180559945309f));          // EM.cpp:195
        sumr_acc += stat_round_sumr((double)one_minus_q * (double)invZ);                 // 1 - sum_i r[i]  (EM.cpp:509-513)
        seq_cnt++;

        if (WRITE_R) {                                       // EM::getR layout: r[L-W-i], i = p-W+1
            float* ro = a.r_out + (a.sv.pos_off[seq] - a.r_base);
#pragma unroll
            for (int m = 0; m < M; m++) {
                const uint32_t p = p0 + m;
                if (p < L) ro[L - 1u - p] = U[m];
            }
        }

        if (ACCUM) {
            // ---- M-step (EM.cpp:236-242): the adjoint chain, wide groups first (they are the motif's last)
            __builtin_amdgcn_s_setprio(3);
            unsigned long long F[M], nz[M];
            uint32_t rad6[M], rad5[M];
            const uint32_t c5 = lds_offset(cnt5), c6 = lds_offset(cnt6);
#pragma unroll
            for (int m = 0; m < M; m++) {
                F[m] = to_fixed40_pre(U[m]);
                nz[m] = __ballot(F[m] != 0ull);
                rad6[m] = c6 + row6[m] * 8u;
                rad5[m] = c5 + row5[m] * 8u;
            }
            // F is a ring: logical slot m lives in F[(m + off) mod M]; a step of G moves the G slots that arrive
            // from the next lane in place (one DPP pair each) and their non-zero masks with them
            static_for<A>([&](auto wc) {
                constexpr int w = decltype(wc)::value;
                constexpr int off = (4 * w) % M;
#pragma unroll
                for (int m = 0; m < M; m++)
                    lds_add_u64_exec_big<(int)(w * R6T * 8u)>(rad6[m], F[(m + off) % M], nz[(m + off) % M]);
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const int idx = (c + off) % M;
                    F[idx] = wave_shl1_u64(F[idx]);
                    nz[idx] >>= 1;
                }
            });
            static_for<4 * NQ>([&](auto uc) {
                constexpr int u = decltype(uc)::value;
                if ((uint32_t)u < B) {
                    constexpr int off = (4 * A + 3 * u) % M;
#pragma unroll
                    for (int m = 0; m < M; m++)
                        lds_add_u64_exec_big<(int)(u * R5T * 8u)>(rad5[m], F[(m + off) % M], nz[(m + off) % M]);
#pragma unroll
                    for (int c = 0; c < 3; c++) {
                        const int idx = (c + off) % M;
                        F[idx] = wave_shl1_u64(F[idx]);
                        nz[idx] >>= 1;
                    }
                }
            });
            // ---- what the virtual count rows collected (one window per cell) goes to the log
            wave_lds_sync();
            unsigned long long acc = 0ull;
            if (fix) {
                unsigned long long* cell = lane_wide ? cnt6 + (size_t)(T - 1u - lane_t) * R6T + vbase6 + lane_b
                                                     : cnt5 + (size_t)(B - 1u - lane_t) * R5T + vbase5 + lane_b;
                acc = *cell;
                *cell = 0ull;
            }
            // code: group (3 bits), then the y of its up to four columns (7 bits each, >= Y = nothing to log: neutral,
            // beyond the edge, or a resident bin, which takes the sum here)
            // (fields of columns beyond the lane's group hold Y already: yfix is built that way)
            if (acc != 0ull) {
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const uint32_t yc = (yfix >> (7 * c)) & 0x7fu;
                    if (yc < Y && ((resident_fill >> (7 * c + 6)) & 1u) != 0u)   // a resident bin: added here, not logged
                        atomicAdd(&n1p[(lane_col0 + (uint32_t)c) * Y + yc], acc);
                }
            }
            const uint32_t ylog = yfix | resident_fill;          // what is left to log: fields still below Y
            const uint32_t code = lane_t | (ylog << 3);
            if ((~ylog & (Y | (Y << 7) | (Y << 14) | (Y << 21))) == 0u) acc = 0ull;   // nothing left to log (also: a group wholly beyond the edge)
            const unsigned long long nzm = __ballot(acc != 0ull);
            if (acc != 0ull) {
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(nzm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)nzm, 0u));
                GrpLogEntry* const log_now = reinterpret_cast<GrpLogEntry*>(kp->fix_log) + (size_t)(blockIdx.x * WAVES + wave) * kp->fix_log_cap;
                grp_log_store(log_now, nlog + rank, acc, code);
            }
            nlog += (uint32_t)__builtin_popcountll(nzm);
            wave_lds_sync();
        }
    }

    // ---- block epilogue (the statistics take the narrow odds table's place: every wave must be past its last sequence)
    lds_drain();
    BAMM_PHASE(3);                                           // wave 0 is through its sequences
    __syncthreads();
    BAMM_PHASE(4);                                           // ... and the block's slowest wave
    if (lane == 0) {
        stat_lds[wave * 3 + 0] = llh_acc;
        stat_lds[wave * 3 + 1] = (double)seq_cnt - sumr_acc;
        stat_lds[wave * 3 + 2] = (double)seq_cnt;
    }
    __syncthreads();
    if (a.acc == nullptr) return;                            // getR(): responsibilities only
    if constexpr (ACCUM) {
        // logged virtual-row sums -> single-column bins [j][y], which take the wide odds table's place
        for (uint32_t i = threadIdx.x; i < W * Y; i += THREADS) n1[i] = 0ull;
        __syncthreads();
        BAMM_PHASE(7);                                       // statistics parked, bins zeroed
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
                    const uint32_t tt = code[u] & 7u;
                    const uint32_t col0 = tt >= B ? 3u * B + 4u * (tt - B) : 3u * tt;
#pragma unroll
                    for (int c = 0; c < 4; c++) {
                        const uint32_t yc = (code[u] >> (3 + 7 * c)) & 127u;
                        if (yc < Y) atomicAdd(&n1[(col0 + (uint32_t)c) * Y + yc], acc[u]);
                    }
                }
            }
        }
        __syncthreads();
        BAMM_PHASE(5);                                       // the log is folded
        // Marginalisation in units of 16 table rows, every lane of every read instruction at work: a cell of a wide group
        // (64 rows carry its 3-mer) is four units on the four lanes of a quad, summed across the quad with two DPP adds; a
        // narrow cell is one unit; the wide units come first and fill whole rounds of the block, so no wave runs both
        // kinds (one item per cell made every wave issue 16 + 64 reads per round with 60 / 40 % of its lanes: the LDS
        // pipe's 2 us per round).  Every lane starts its sixteen rows elsewhere ((lane + lane / 4) mod 16): at equal
        // steps the lanes' rows differ in digits that do not reach the bank, rotated they spread over all of them.
        // Integer sums: any order gives the same total.  The totals go to the bins and from there out in the
        // accumulator's order (a scattered atomic instruction costs the L2 one operation per cache line it touches).
        constexpr uint32_t kWideUnits = 4u * (uint32_t)A * 4u * Y;           // [w][c][y][quarter]
        const uint32_t narrow_cells = 3u * B * Y;
        const uint32_t rot = ((uint32_t)lane + ((uint32_t)lane >> 2)) & 15u;
        for (uint32_t u = threadIdx.x; u < kWideUnits + narrow_cells; u += THREADS) {
            unsigned long long acc = 0ull;
            uint32_t j, yy;
            bool owner = true;
            unsigned long long v[16];
            if (u < kWideUnits) {                                               // quad-uniform: THREADS and kWideUnits are multiples of 4
                const uint32_t cell = u >> 2, quarter = u & 3u;
                yy = cell & (Y - 1u);
                const uint32_t wc = cell >> 6, w = wc >> 2, c = wc & 3u;
                j = 3u * B + 4u * w + c;
                const unsigned long long* tab = cnt6 + (size_t)((uint32_t)A - 1u - w) * R6T;
                const uint32_t lowd = 2u * (3u - c), lmask = (1u << lowd) - 1u;
#pragma unroll
                for (uint32_t rr = 0; rr < 16u; rr++) {
                    const uint32_t r = quarter * 16u + ((rr + rot) & 15u), h = r >> lowd, l = r & lmask;
                    v[rr] = tab[((((h << 6) | yy) << lowd) | l) & 4095u];
                }
#pragma unroll
                for (uint32_t rr = 0; rr < 16u; rr++) acc += v[rr];
                acc += quad_perm_u64<0xB1>(acc);                                 // lanes 1,0,3,2
                acc += quad_perm_u64<0x4E>(acc);                                 // lanes 2,3,0,1
                owner = quarter == 0u;
            } else {
                const uint32_t cell = u - kWideUnits;
                yy = cell & (Y - 1u);
                j = cell >> 6;
                const uint32_t tt = j / 3u, c = j - 3u * tt;
                const unsigned long long* tab = cnt5 + (size_t)(B - 1u - tt) * R5T;
                const uint32_t lowd = 2u * (2u - c), lmask = (1u << lowd) - 1u;
#pragma unroll
                for (uint32_t rr = 0; rr < 16u; rr++) {
                    const uint32_t r = (rr + rot) & 15u, h = r >> lowd, l = r & lmask;
                    v[rr] = tab[((((h << 6) | yy) << lowd) | l) & 1023u];
                }
#pragma unroll
                for (uint32_t rr = 0; rr < 16u; rr++) acc += v[rr];
            }
            if (owner) n1[j * Y + yy] += acc + (j < n1c ? n1p[j * Y + yy] : 0ull);     // the cell's total, in its bin
        }
        BAMM_PHASE(13);
        __syncthreads();
        BAMM_PHASE(14);
        for (uint32_t o = threadIdx.x; o < W * Y; o += THREADS) {               // o = y*W + j: consecutive global cells
            const uint32_t yy = o / W, j = o - yy * W;
            const unsigned long long acc = n1[j * Y + yy];
            if (acc) acc_add(a.acc + o, (long long)acc);
        }
    }
    if (threadIdx.x < 3) {
        double acc = 0.0;
        for (uint32_t w = 0; w < WAVES; w++) acc += stat_lds[w * 3 + threadIdx.x];
        acc_add_stat(a.acc, W * Y, threadIdx.x, acc);
    }
    BAMM_PHASE(6);                                           // marginalised, atomics issued (thread 0's)
    if constexpr (ACCUM) {
        // in-kernel all-reduce: the last block to get here sums the GPUs' totals.  Its arguments are read from the
        // kernel-argument segment HERE (opaque pointer: the scalar loads cannot be hoisted above the sequence loop, where
        // every live SGPR is one more spill -- 44 -> 100 spilled SGPRs and +3 % on the pass when they were ordinary arguments)
        const __attribute__((address_space(4))) GrpKernelArgs* kq =
            (const __attribute__((address_space(4))) GrpKernelArgs*)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(kq));
        if (kq->peer.world > 1u) {
            __syncthreads();                                 // the LDS is nobody's any more
            peer_allreduce_tail(&kq->peer, kq->e.acc, reinterpret_cast<uint32_t*>(lds_raw));
        }
    }
}

template <int M, int A, int NQ, int THREADS>
int launch_mix_variant(bool accum, bool write_r, const GrpKernelArgs& a, uint32_t blocks, hipStream_t st) {
    const size_t lds = a.g.lds_bytes;
    int rc;
    if (write_r) {
        if ((rc = allow_lds(reinterpret_cast<const void*>(&k_em_mix<M, A, NQ, false, true, THREADS>), lds))) return rc;
        if (blocks == kPrimeOnly) return prime_kernel(reinterpret_cast<const void*>(&k_em_mix<M, A, NQ, false, true, THREADS>));
        hipLaunchKernelGGL((k_em_mix<M, A, NQ, false, true, THREADS>), dim3(blocks), dim3(THREADS), lds, st, a);
    } else if (accum) {
        if ((rc = allow_lds(reinterpret_cast<const void*>(&k_em_mix<M, A, NQ, true, false, THREADS>), lds))) return rc;
        if (blocks == kPrimeOnly) return prime_kernel(reinterpret_cast<const void*>(&k_em_mix<M, A, NQ, true, false, THREADS>));
        hipLaunchKernelGGL((k_em_mix<M, A, NQ, true, false, THREADS>), dim3(blocks), dim3(THREADS), lds, st, a);
    } else {
        if ((rc = allow_lds(reinterpret_cast<const void*>(&k_em_mix<M, A, NQ, false, false, THREADS>), lds))) return rc;
        if (blocks == kPrimeOnly) return prime_kernel(reinterpret_cast<const void*>(&k_em_mix<M, A, NQ, false, false, THREADS>));
        hipLaunchKernelGGL((k_em_mix<M, A, NQ, false, false, THREADS>), dim3(blocks), dim3(THREADS), lds, st, a);
    }
    return BAMM_OK;
}

}  // namespace
}  // namespace bamm
