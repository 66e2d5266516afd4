// gfx950 (MI355X / CDNA4) kernels of the BaMMmotif2 EM hot path.
//
// What they compute (reference lines, relative to /root/reference/src):
//   k_em_seq        EM::EStep  refinement/EM.cpp:149-196   (responsibilities, log-likelihood)
//                   EM::MStep  refinement/EM.cpp:231-243   (order-K fractional counts)
//                   EM::optimize_q's sum over r            refinement/EM.cpp:509-513
//   (the model update, the odds table and the reductions: model.hip)
//   k_score         ScoreSeqSet::calcLogOdds seq_scoring/ScoreSeqSet.cpp:41-66
//
//   k_e_slice / k_m_slice  the same two chains cut into column ranges for tables beyond one
//                   CU's LDS (k >= 4), chain state and r in HBM
//
// Design: one 64-lane wavefront owns one sequence at a time.  Lane l
// holds M consecutive positions p = l*M+m in registers.  The odds table lives in LDS as
// [W/4][Y+1][4] (row Y and padding columns = 1.0f); window products are a systolic chain
//       U_j(p) = U_{j-1}(p-1) * s[j][y(p)]
// whose only cross-lane traffic is ONE `wave_shr:1` DPP move per motif column; a position's
// four columns arrive with one hand-issued ds_read_b128.  The M-step is the adjoint chain on
// 64-bit fixed-point addends (ds_add_u64 into n[j][y][copy]; ds_add_f32 is ~25x slower on
// gfx950): dense form = register ring shifted with `wave_shl:1`, predicated by precomputed
// wave masks; sparse form = the non-zero windows compacted into a per-wave LDS list.  Nothing
// but the 2-bit sequence stream (+ the N exceptions) is read from HBM; per block one partial
// table is written.  No MFMA: this is gather/scatter, not a contraction.

#include "device_utils.h"

#include <algorithm>
#include <cfloat>

namespace bamm {

const int kMClasses[kNumMClasses] = {1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 14, 16, 20, 24, 28, 32, 40, 48, 56, 64, 80, 96, 128};

int m_class_for_len(uint32_t L) {
    for (int i = 0; i < kNumMClasses; i++)
        if (L <= 64u * (uint32_t)kMClasses[i]) return i;
    return -1;
}

uint32_t max_threads_for_mclass(int mclass) {
    int M = kMClasses[mclass];
    return M <= 8 ? 1024u : (M <= 20 ? 768u : (M <= 64 ? 512u : 256u));   // 4 / 3 / 2 / 1 waves per SIMD: what the VGPR use of the class allows
}

namespace {


// ---- dense M-step walk over `ncols` columns, last column first ---------------------------------
// F is kept as a ring: after t shifts logical slot m lives in F[(m+t) mod M]; the shift itself is one
// in-place DPP pair on F[t] (the value leaving becomes the value arriving from the next lane), so no
// register moves.  The ring index must be a compile-time value, i.e. the walk is unrolled: up to 32
// positions per lane over a whole turn (M steps, M*M adds); beyond that a turn would be thousands of
// adds, the compiler gives up on the unrolling and the arrays land in scratch memory (5x slower).
// Those classes unroll 8 steps and then turn the ring back by 8 places (2*M register moves per 8*M adds).
template <int M>
__device__ __forceinline__ void dense_count_walk(uint32_t ncols, uint32_t col, uint32_t stride, const uint32_t (&ya)[M],
                                                 unsigned long long (&F)[M], unsigned long long (&nz)[M],
                                                 const unsigned long long (&padm)[M], const uint32_t (&y)[M], uint32_t Y) {
    constexpr int TB = M <= 32 ? M : 8;
    for (uint32_t jb = 0; jb < ncols; jb += TB) {
#pragma unroll
        for (int t = 0; t < TB; t++) {
            if (jb + t < ncols) {
#pragma unroll
                for (int m = 0; m < M; m++)
                    lds_add_slot<M>(col + ya[m], F[(m + t) % M], padm[m], nz[(m + t) % M], y[m] != Y);
                F[t] = wave_shl1_u64(F[t]);
                nz[t] = __ballot(F[t] != 0ull);
                col -= stride;
            }
        }
        if constexpr (TB < M) {
            unsigned long long Fr[M], nr[M];
#pragma unroll
            for (int m = 0; m < M; m++) { Fr[m] = F[(m + TB) % M]; nr[m] = nz[(m + TB) % M]; }
#pragma unroll
            for (int m = 0; m < M; m++) { F[m] = Fr[m]; nz[m] = nr[m]; }
        }
    }
}

// ---- fused E+M sequence kernel --------------------------------------------------------------
template <int M, bool ACCUM, bool WRITE_R, int THREADS>
__global__ void __launch_bounds__(THREADS) k_em_seq(EmKernelArgs a) {
    if (a.stop != nullptr && *a.stop != 0u) return;      // optimize(): the stop rule fired in an earlier pass
    if (a.nnz_prev != nullptr && ((*a.nnz_prev > a.nnz_limit) != (a.run_if_long != 0u))) return;   // the pass's other flavour runs

    extern __shared__ __align__(16) float lds[];
    const uint32_t W = a.W, Y = a.Y, Ys = a.Y + 1u;
    const uint32_t Wq = (W + 3u) >> 2;
    const uint32_t pad = 4u * Wq - W;
    // odds table as [W/4][Y+1][4]: one ds_read_b128 fetches the row of FOUR motif columns for a
    // position (2.3 LDS cycles per column instead of 4.2 for ds_read_b32, tools/lds_bench2.hip);
    // row Y and the padding columns are 1.0f
    float* s_lds = lds;
    const uint32_t n_off = Wq * Ys * 4u;                 // 16-byte aligned by construction
    // [W][Y+1][C] 64-bit fixed point (2^-40 units); C = 2^logC private copies, copy = lane mod C,
    // cut the same-address serialisation of ds_add_u64 (tools/lds_bench2.hip: 5.6 -> 2.7 ns).
    // Row Y of every column is never written (positions beyond LW1 are exec-masked off).
    unsigned long long* n_lds = reinterpret_cast<unsigned long long*>(lds + n_off);
    const uint32_t logC = ACCUM ? a.logC : 0u;
    double* stat_lds = reinterpret_cast<double*>(lds + n_off + (ACCUM ? (2u * W * Ys) << logC : 0u));  // [waves][3]

    // the wave's first sequence is on its way from HBM while the block builds its tables
    const int lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t waves_per_block = blockDim.x >> 6;
    const uint32_t total_waves = gridDim.x * waves_per_block;
    uint32_t t = blockIdx.x * waves_per_block + wave;
    RawSeq<M> nxt{};
    if (t < a.sv.count) nxt = fetch_seq<M>(a.sv, t, lane);

    for (uint32_t i = threadIdx.x; i < Wq * Ys * 4u; i += blockDim.x) {
        const uint32_t jq = i / (Ys * 4u), rem = i - jq * Ys * 4u, yy = rem >> 2, j = jq * 4u + (rem & 3u);
        // the 4*Wq - W padding columns sit in FRONT of column 0 and hold 1.0f: every quad then takes all
        // four chain steps without a guard (a guard between two steps costs M register moves at the
        // merge point), and slot p still ends up with window p-(W-1)
        s_lds[i] = (j >= pad) ? a.s[(size_t)(j - pad) * Ys + yy] : 1.0f;
    }
    if (ACCUM)
        for (uint32_t i = threadIdx.x; i < (W * Ys) << logC; i += blockDim.x) n_lds[i] = 0ull;
    __syncthreads();

    const float q = *a.q;
    const float one_minus_q = 1.0f - q;

    double llh_acc = 0.0, sumr_acc = 0.0;
    uint32_t seq_cnt = 0;
    [[maybe_unused]] unsigned long long nnz_acc = 0ull;   // windows with a non-zero addend (sliced path: next pass's lists-or-dense choice)

    for (; t < a.sv.count; t += total_waves) {
        const RawSeq<M> cur = nxt;
        if (t + total_waves < a.sv.count) nxt = fetch_seq<M>(a.sv, t + total_waves, lane);   // prefetch
        const uint32_t seq = cur.seq;
        if (WRITE_R && (seq < a.seq_begin || seq >= a.seq_end)) continue;
        if (!cur.ok) continue;
        const uint32_t L = cur.L;
        const uint32_t LW1 = L - W + 1u;
        const uint32_t p0 = (uint32_t)lane * M;

        // wave priorities by phase, as in k_em_grp (grouped.hip): the LDS-bound M-step ahead of the
        // VALU-bound E-step in the issue arbitration
        __builtin_amdgcn_s_setprio(0);
        uint32_t y[M];
        // EM.cpp:167: only positions ij < LW1 take part; everything beyond reads the neutral row
        decode_raw<M>(cur, a.sv, Y, LW1, lane, y);
        __builtin_amdgcn_s_setprio(2);

        // ---- E-step: U[m] after column j = prod_{j'<=j} s[j'][y(p-j+j')]  (EM.cpp:167-176)
        float U[M];
        if constexpr (M > 16) {
            // long-sequence classes: 4*M live floats per quad would spill (256 VGPRs); read the
            // same [W/4][Y+1][4] table one column at a time instead
#pragma unroll
            for (int m = 0; m < M; m++) U[m] = s_lds[(size_t)(pad >> 2) * Ys * 4u + (pad & 3u) + y[m] * 4u];
            for (uint32_t j = 1; j < W; j++) {
                const float* sj = s_lds + (size_t)((j + pad) >> 2) * Ys * 4u + ((j + pad) & 3u);
                const float u0 = mul_wave_shr1(sj[y[0] * 4u], U[M - 1]);    // one v_mul_f32_dpp: lane 0 multiplies by 1
#pragma unroll
                for (int m = M - 1; m >= 1; m--) U[m] = U[m - 1] * sj[y[m] * 4u];
                U[0] = u0;
            }
        } else {
        uint32_t sa[M];                                  // LDS byte address of row y(p) in the current quad
        const uint32_t s_base = lds_offset(s_lds);
#pragma unroll
        for (int m = 0; m < M; m++) sa[m] = s_base + y[m] * 16u;
#define BAMM_SEQ_STEP(COMP)                                                     \
            { const float c = mul_wave_shr1(sv[0].COMP, U[M - 1]);              \
              _Pragma("unroll") for (int m = M - 1; m >= 1; m--) U[m] = U[m - 1] * sv[m].COMP; \
              U[0] = c; }
        {                                                // first quad: its column 0 starts the chain
            f32x4 sv[M];
#pragma unroll
            for (int m = 0; m < M; m++) sv[m] = lds_read_b128(sa[m]);
#pragma unroll
            for (int m = 0; m < M; m++) sa[m] += Ys * 16u;
            lds_wait(sv);
#pragma unroll
            for (int m = 0; m < M; m++) U[m] = sv[m].x;
            BAMM_SEQ_STEP(y)
            BAMM_SEQ_STEP(z)
            BAMM_SEQ_STEP(w)
        }
        for (uint32_t jq = 1; jq < Wq; jq++) {
            f32x4 sv[M];
#pragma unroll
            for (int m = 0; m < M; m++) sv[m] = lds_read_b128(sa[m]);
#pragma unroll
            for (int m = 0; m < M; m++) sa[m] += Ys * 16u;
            lds_wait(sv);
            BAMM_SEQ_STEP(x)
            BAMM_SEQ_STEP(y)
            BAMM_SEQ_STEP(z)
            BAMM_SEQ_STEP(w)
        }
#undef BAMM_SEQ_STEP
        }
        // slot p now holds the product of window start i = p-(W-1); valid for W-1 <= p < L
        __builtin_amdgcn_s_setprio(1);
        const float pos_i = q / (float)LW1;              // EM.cpp:160
        float zpart = 0.0f;
#pragma unroll
        for (int m = 0; m < M; m++) {
            const uint32_t p = p0 + m;
            const bool valid = (p + 1u >= W) && (p < L);
            U[m] = valid ? U[m] * pos_i : 0.0f;          // EM.cpp:180
            zpart += U[m];
        }
        const float Z = one_minus_q + wave_sum(zpart);   // EM.cpp:154,181
        const float invZ = 1.0f / Z;                     // one IEEE division per sequence
        // the M-step wants r in units of the count accumulator: the power-of-two scale rides on 1/Z (exact)
        const float invZs = ACCUM ? invZ * a.fix_scale : invZ;
#pragma unroll
        for (int m = 0; m < M; m++) U[m] = U[m] * invZs; // EM.cpp:185-187 (r/Z within 1 ulp)
        llh_acc += stat_round_llh((double)logf(Z));      // EM.cpp:195
        sumr_acc += 1.0 - stat_round_sumr((double)one_minus_q / (double)Z);  // = sum_i r[i]  (EM.cpp:509-513)
        seq_cnt++;

        if (WRITE_R && a.list_r != nullptr) {
            // sliced path: only the windows whose fixed-point addend is non-zero (r * scale >= 2^-40, the very
            // test the M-step applies) are worth a trip through HBM: once the model is informative that is
            // a quarter of them.  Per sequence: responsibilities and slots, compacted in lane order.
            const uint64_t base = a.sv.pos_off[seq];
            uint32_t nnz = 0;
#pragma unroll
            for (int m = 0; m < M; m++) {
                const bool nzm = U[m] * a.fix_scale >= 0x1p-40f;
                const unsigned long long mask = __ballot(nzm);
                const uint32_t at = nnz + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
                if (nzm) {
                    a.list_r[base + at] = U[m];
                    a.list_p[base + at] = (uint16_t)(p0 + (uint32_t)m);
                }
                nnz += (uint32_t)__popcll(mask);
            }
            if (lane == 0) a.list_n[seq] = nnz;
            nnz_acc += nnz;
        } else if (WRITE_R) {                            // EM::getR layout: r[L-W-i], i = p-W+1
            if (a.nnz_out != nullptr) {
#pragma unroll
                for (int m = 0; m < M; m++) nnz_acc += (uint32_t)__popcll(__ballot(U[m] * a.fix_scale >= 0x1p-40f));
            }
            float* ro = a.r_out + (a.sv.pos_off[seq] - a.r_base);
            bool done = false;
            if constexpr (M % 4 == 0) {
                // a lane's M slots are M consecutive floats (descending): four at a time, 16-byte stores
                // at 4-byte alignment, instead of M scattered dwords per lane
                if (p0 + (uint32_t)M <= L) {
#pragma unroll
                    for (int m = 0; m < M; m += 4) {
                        f32x4u v;
                        v.x = U[m + 3]; v.y = U[m + 2]; v.z = U[m + 1]; v.w = U[m];
                        *reinterpret_cast<f32x4u*>(ro + (L - 1u - (p0 + (uint32_t)m + 3u))) = v;
                    }
                    done = true;
                }
            }
            if (!done) {
#pragma unroll
                for (int m = 0; m < M; m++) {
                    const uint32_t p = p0 + m;
                    if (p < L) ro[L - 1u - p] = U[m];
                }
            }
        }

        if (ACCUM) {
            __builtin_amdgcn_s_setprio(3);
            // ---- M-step (EM.cpp:236-242): position p, column j receives r(i = p-j), which
            // sits in slot p+(W-1-j): walk j downwards and shift the slots one step per column.
            // Counts are accumulated as 64-bit fixed point (2^-40 units) with ds_add_u64: LDS
            // float atomics (ds_add_f32) run ~25x slower on gfx950 (tools/lds_bench.hip), and
            // integer sums are exact, so the result does not depend on scheduling order.
            unsigned long long F[M];
            const uint32_t copy = (uint32_t)lane & ((1u << logC) - 1u);
            const uint32_t stride = (Ys << logC) * 8u;
#pragma unroll
            for (int m = 0; m < M; m++) F[m] = to_fixed40(U[m]);

            // ---- sparse path: once the model is informative most windows have r < 2^-41, i.e. an
            // addend of exactly 0.  The non-zero windows are compacted into a per-wave list and each
            // lane then walks the W columns of ~nnz/64 windows: nnz*W/64 dense adds instead of M*W
            // mostly-idle ones.  Identical sums (integer adds commute), chosen per sequence.
            bool dense = true;
            if (a.sparse_cap != 0u) {
                uint32_t nnz = 0, lpos[M];
#pragma unroll
                for (int m = 0; m < M; m++) {
                    const unsigned long long mask = __ballot(F[m] != 0ull);
                    lpos[m] = nnz + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
                    nnz += (uint32_t)__popcll(mask);
                }
                if (nnz <= a.sparse_cap) {
                    dense = false;
                    const uint32_t cap = a.sparse_cap;
                    unsigned long long* listF = reinterpret_cast<unsigned long long*>(
                        reinterpret_cast<char*>(stat_lds) + 16u * 3u * sizeof(double) + (size_t)wave * a.sparse_wave_bytes);
                    unsigned short* listP = reinterpret_cast<unsigned short*>(listF + cap);
                    unsigned short* ybuf = listP + cap;
#pragma unroll
                    for (int m = 0; m < M; m++) {
                        if (F[m] != 0ull) { listF[lpos[m]] = F[m]; listP[lpos[m]] = (unsigned short)(p0 + m); }
                        ybuf[p0 + m] = (unsigned short)y[m];
                    }
                    const uint32_t ecnt = (nnz + 63u) >> 6;
                    for (uint32_t e = 0; e < ecnt; e++) {
                        const uint32_t idx = e * 64u + (uint32_t)lane;
                        const bool ok = idx < nnz;
                        const unsigned long long Fe = ok ? listF[idx] : 0ull;
                        uint32_t q = ok ? (uint32_t)listP[idx] - (W - 1u) : 0u;     // window start i
                        unsigned long long* ncol = n_lds + copy;
                        // eight columns' y first, then their adds: one LDS round trip per batch instead of
                        // one per column (row Y of a column is never read: adds that land there are dropped)
                        const uint32_t cstride = Ys << logC;
                        if (ok) {
                            for (uint32_t jb = 0; jb < W; jb += 8u, q += 8u, ncol += 8u * cstride) {
                                uint32_t yy[8];
#pragma unroll
                                for (int u = 0; u < 8; u++) yy[u] = (jb + u < W) ? (uint32_t)ybuf[q + u] : Y;
#pragma unroll
                                for (int u = 0; u < 8; u++)
                                    if (jb + u < W) atomicAdd(&ncol[(size_t)u * cstride + (yy[u] << logC)], Fe);
                            }
                        }
                    }
                }
            }
            if (dense) {
            unsigned long long nz[M], padm[M];
            uint32_t ya[M];                              // byte offset of (row y, private copy) inside a column
#pragma unroll
            for (int m = 0; m < M; m++) {
                nz[m] = __ballot(F[m] != 0ull);          // adding an exact 0 is a no-op: those lanes sit out
                padm[m] = __ballot(y[m] != Y);           // positions beyond LW1 take no part (EM.cpp:236)
                ya[m] = ((y[m] << logC) + copy) * 8u;
            }
            dense_count_walk<M>(W, lds_offset(n_lds) + (W - 1u) * stride, stride, ya, F, nz, padm, y, Y);
            }
        }
    }

    // ---- block epilogue: this block's table and statistics into the pass's accumulator
    lds_drain();
    if constexpr (WRITE_R)
        if (a.nnz_out != nullptr && lane == 0 && nnz_acc != 0ull) (void)__hip_atomic_fetch_add(a.nnz_out, nnz_acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (lane == 0) {
        stat_lds[wave * 3 + 0] = llh_acc;
        stat_lds[wave * 3 + 1] = sumr_acc;
        stat_lds[wave * 3 + 2] = (double)seq_cnt;
    }
    __syncthreads();
    if (a.acc == nullptr) return;                        // getR(): responsibilities only
    if (ACCUM) {
        for (uint32_t o = threadIdx.x; o < W * Y; o += blockDim.x) {       // o = y*W + j: consecutive global cells
            const uint32_t yy = o / W, j = o - yy * W;
            unsigned long long acc = 0ull;
            for (uint32_t c = 0; c < (1u << logC); c++) acc += n_lds[(((size_t)j * Ys + yy) << logC) + c];
            if (acc) acc_add(a.acc + o, (long long)acc);
        }
    }
    if (threadIdx.x < 3) {
        double acc = 0.0;
        for (uint32_t w = 0; w < waves_per_block; w++) acc += stat_lds[w * 3 + threadIdx.x];
        acc_add_stat(a.acc, W * Y, threadIdx.x, acc);
    }
}


// ---- column-sliced path (tables that do not fit the fused kernel's LDS budget, e.g. k=4) -----
// The same two systolic chains, cut into column ranges [j0,j1).  The E-chain state (one float
// per position slot) travels through HBM between slices; the last E-slice normalises and leaves
// the responsibilities r(slot) in that buffer.  An M-slice needs no carried state: at column j
// slot p holds r(p + W-1-j), which it reads straight from the buffer.
template <int M, int THREADS>
__global__ void __launch_bounds__(THREADS) k_e_slice(EmKernelArgs a, uint32_t j0, uint32_t j1, int last) {
    if (a.stop != nullptr && *a.stop != 0u) return;      // optimize(): the stop rule fired in an earlier pass

    extern __shared__ float lds[];
    const uint32_t W = a.W, Y = a.Y, Ys = a.Y + 1u, nc = j1 - j0;
    float* s_lds = lds;                                               // [nc][Y+1]
    double* stat_lds = reinterpret_cast<double*>(lds + ((nc * Ys + 1u) & ~1u));
    for (uint32_t i = threadIdx.x; i < nc * Ys; i += blockDim.x) s_lds[i] = a.s[(size_t)j0 * Ys + i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t waves_per_block = blockDim.x >> 6;
    const uint32_t total_waves = gridDim.x * waves_per_block;
    const float q = *a.q;
    const float one_minus_q = 1.0f - q;
    double llh_acc = 0.0, sumr_acc = 0.0;
    uint32_t seq_cnt = 0;
    for (uint32_t t = blockIdx.x * waves_per_block + wave; t < a.sv.count; t += total_waves) {
        const uint32_t seq = pick_sequence(a.sv, t);
        if (a.sv.mask && !a.sv.mask[seq]) continue;
        const uint32_t L = a.sv.len[seq];
        const uint32_t LW1 = L - W + 1u;
        const uint32_t p0 = (uint32_t)lane * M;
        float* state = a.r_out + a.sv.pos_off[seq];
        uint32_t y[M];
        decode_positions<M>(a.sv, seq, L, Y, LW1, lane, y);
        float U[M];
        const float* sj = s_lds;
        uint32_t j = j0;
        if (j0 == 0) {
#pragma unroll
            for (int m = 0; m < M; m++) U[m] = sj[y[m]];
            j = 1;
            sj += Ys;
        } else {
#pragma unroll
            for (int m = 0; m < M; m++) U[m] = (p0 + m < L) ? state[p0 + m] : 1.0f;
        }
        for (; j < j1; j++, sj += Ys) {
            const float u0 = mul_wave_shr1(sj[y[0]], U[M - 1]);
#pragma unroll
            for (int m = M - 1; m >= 1; m--) U[m] = U[m - 1] * sj[y[m]];
            U[0] = u0;
        }
        if (last) {
            const float pos_i = q / (float)LW1;
            float zpart = 0.0f;
#pragma unroll
            for (int m = 0; m < M; m++) {
                const uint32_t p = p0 + m;
                const bool valid = (p + 1u >= W) && (p < L);
                U[m] = valid ? U[m] * pos_i : 0.0f;
                zpart += U[m];
            }
            const float Z = one_minus_q + wave_sum(zpart);
            const float invZ = 1.0f / Z;
#pragma unroll
            for (int m = 0; m < M; m++) U[m] = U[m] * invZ;
            llh_acc += stat_round_llh((double)logf(Z));
            sumr_acc += 1.0 - stat_round_sumr((double)one_minus_q / (double)Z);
            seq_cnt++;
        }
#pragma unroll
        for (int m = 0; m < M; m++)
            if (p0 + m < L) state[p0 + m] = U[m];
    }
    if (lane == 0) {
        stat_lds[wave * 3 + 0] = llh_acc;
        stat_lds[wave * 3 + 1] = sumr_acc;
        stat_lds[wave * 3 + 2] = (double)seq_cnt;
    }
    __syncthreads();
    if (last && threadIdx.x < 3 && a.acc) {
        double acc = 0.0;
        for (uint32_t w = 0; w < waves_per_block; w++) acc += stat_lds[w * 3 + threadIdx.x];
        acc_add_stat(a.acc, W * Y, threadIdx.x, acc);
    }
}

// r_reversed: the E pass left r in the reference's layout r[L-1-slot] (k_em_seq WRITE_R) instead of
// per slot.  sparse_cap: non-zero windows of a sequence compacted into a per-wave list, as in
// k_em_seq (a ds_add_u64 costs the same LDS cycles whether 3 or 64 lanes take part).
template <int M, int THREADS>
__global__ void __launch_bounds__(THREADS) k_m_slice(EmKernelArgs a, uint32_t j0, uint32_t j1, int r_reversed) {
    if (a.stop != nullptr && *a.stop != 0u) return;      // optimize(): the stop rule fired in an earlier pass
    if (a.nnz_prev != nullptr && ((*a.nnz_prev > a.nnz_limit) != (a.run_if_long != 0u))) return;   // the pass's other flavour runs

    extern __shared__ float lds[];
    const uint32_t W = a.W, Y = a.Y, Ys = a.Y + 1u, nc = j1 - j0, logC = a.logC;
    unsigned long long* n_lds = reinterpret_cast<unsigned long long*>(lds);      // [nc][Y+1][C], row Y = dump
    for (uint32_t i = threadIdx.x; i < (nc * Ys) << logC; i += blockDim.x) n_lds[i] = 0ull;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t waves_per_block = blockDim.x >> 6;
    const uint32_t total_waves = gridDim.x * waves_per_block;
    const uint32_t copy = (uint32_t)lane & ((1u << logC) - 1u);
    unsigned char* wscratch = reinterpret_cast<unsigned char*>(n_lds + ((size_t)(nc * Ys) << logC)) + (size_t)wave * a.sparse_wave_bytes;
    for (uint32_t t = blockIdx.x * waves_per_block + wave; t < a.sv.count; t += total_waves) {
        const uint32_t seq = pick_sequence(a.sv, t);
        if (a.sv.mask && !a.sv.mask[seq]) continue;
        const uint32_t L = a.sv.len[seq];
        const uint32_t LW1 = L - W + 1u;
        const uint32_t p0 = (uint32_t)lane * M;
        const float* rs = a.r_out + a.sv.pos_off[seq];
        uint32_t y[M];
        decode_positions<M>(a.sv, seq, L, Y, LW1, lane, y);
        const uint32_t shift = W - j1;                               // slot offset at column j1-1
        unsigned long long F[M];
        bool loaded = false;
        if constexpr (M % 4 == 0) {
            if (r_reversed && p0 + (uint32_t)M + shift <= L) {     // the lane's M values in 16-byte loads
#pragma unroll
                for (int m = 0; m < M; m += 4) {
                    const f32x4u v = *reinterpret_cast<const f32x4u*>(rs + (L - 1u - (p0 + (uint32_t)m + 3u + shift)));
                    F[m] = to_fixed40(v.w * a.fix_scale); F[m + 1] = to_fixed40(v.z * a.fix_scale);
                    F[m + 2] = to_fixed40(v.y * a.fix_scale); F[m + 3] = to_fixed40(v.x * a.fix_scale);
                }
                loaded = true;
            }
        }
        if (!loaded) {
#pragma unroll
            for (int m = 0; m < M; m++) {
                const uint32_t slot = p0 + m + shift;
                F[m] = to_fixed40(slot < L ? rs[r_reversed ? L - 1u - slot : slot] * a.fix_scale : 0.0f);
            }
        }
        bool dense = true;
        if (a.sparse_cap != 0u) {
            const uint32_t cap = a.sparse_cap;
            unsigned long long* list = reinterpret_cast<unsigned long long*>(wscratch);   // addend | position << 48
            unsigned short* ybuf = reinterpret_cast<unsigned short*>(list + cap);         // [64*M]
            uint32_t nnz = 0;
#pragma unroll
            for (int m = 0; m < M; m++) {
                const bool nzm = F[m] != 0ull;
                const unsigned long long mask = __ballot(nzm);
                const uint32_t at = nnz + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
                if (nzm && at < cap) list[at] = F[m] | ((unsigned long long)(p0 + m) << 48);
                nnz += (uint32_t)__popcll(mask);
            }
            if (nnz <= cap) {
                dense = false;
#pragma unroll
                for (int m = 0; m < M; m++) ybuf[p0 + m] = (unsigned short)y[m];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const uint32_t ecnt = (nnz + 63u) >> 6;
                for (uint32_t e = 0; e < ecnt; e++) {
                    const uint32_t idx = e * 64u + (uint32_t)lane;
                    if (idx < nnz) {
                        const unsigned long long ent = list[idx];
                        const unsigned long long Fe = ent & 0xffffffffffffull;
                        // the entry sits at position q in column j1-1: it is r of the window that
                        // started at q - (j1-1); that window's column j is at q - (j1-1) + j
                        uint32_t q = (uint32_t)(ent >> 48) + j0 + 1u - j1;
                        unsigned long long* ncol = n_lds + copy;
                        // eight columns' y first, then their adds: one LDS round trip per batch instead of
                        // one per column (row Y of every column is a dump nobody reads: no check)
                        const uint32_t cstride = Ys << logC;
                        for (uint32_t jb = j0; jb < j1; jb += 8u, q += 8u, ncol += 8u * cstride) {
                            uint32_t yy[8];
#pragma unroll
                            for (int u = 0; u < 8; u++) yy[u] = (jb + u < j1) ? (uint32_t)ybuf[q + u] : Y;
#pragma unroll
                            for (int u = 0; u < 8; u++)
                                if (jb + u < j1) atomicAdd(&ncol[(size_t)u * cstride + (yy[u] << logC)], Fe);
                        }
                    }
                }
            }
        }
        if (dense) {
            unsigned long long nz[M], padm[M];
            uint32_t ya[M];
#pragma unroll
            for (int m = 0; m < M; m++) {
                nz[m] = __ballot(F[m] != 0ull);
                padm[m] = __ballot(y[m] != Y);
                ya[m] = ((y[m] << logC) + copy) * 8u;
            }
            const uint32_t stride = (Ys << logC) * 8u;
            dense_count_walk<M>(nc, lds_offset(n_lds) + (nc - 1u) * stride, stride, ya, F, nz, padm, y, Y);
        }
    }
    lds_drain();
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < nc * Y; i += blockDim.x) {          // i = y*nc + j: runs of nc consecutive cells
        const uint32_t yy = i / nc, j = i - yy * nc;
        unsigned long long acc = 0ull;
        for (uint32_t c = 0; c < (1u << logC); c++) acc += n_lds[(((size_t)j * Ys + yy) << logC) + c];
        if (acc) acc_add(a.acc + (size_t)yy * W + j0 + j, (long long)acc);
    }
}

// ---- M-slice over the E pass's compacted lists (sliced path, whole odds table in LDS) -----------------
// Every lane takes listed windows of the wave's sequence and walks the slice's columns: y from a per-wave
// copy of the decoded sequence, one ds_add_u64 per (window, column).  The same integer addends as the dense
// walk of k_m_slice, so the counts are bit-identical; nothing is compacted or converted here any more.
template <int M, int THREADS>
__global__ void __launch_bounds__(THREADS) k_m_list(EmKernelArgs a, uint32_t j0, uint32_t j1) {
    if (a.stop != nullptr && *a.stop != 0u) return;      // optimize(): the stop rule fired in an earlier pass
    if (a.nnz_prev != nullptr && ((*a.nnz_prev > a.nnz_limit) != (a.run_if_long != 0u))) return;   // the pass's other flavour runs

    extern __shared__ float lds[];
    const uint32_t W = a.W, Y = a.Y, Ys = a.Y + 1u, nc = j1 - j0, logC = a.logC;
    unsigned long long* n_lds = reinterpret_cast<unsigned long long*>(lds);      // [nc][Y+1][C], row Y = dump
    for (uint32_t i = threadIdx.x; i < (nc * Ys) << logC; i += blockDim.x) n_lds[i] = 0ull;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t waves_per_block = blockDim.x >> 6;
    const uint32_t total_waves = gridDim.x * waves_per_block;
    const uint32_t copy = (uint32_t)lane & ((1u << logC) - 1u);
    unsigned short* ybuf = reinterpret_cast<unsigned short*>(n_lds + ((size_t)(nc * Ys) << logC)) + (size_t)wave * (64u * M + 8u);
    const uint32_t cstride = Ys << logC;
    // A wave's next sequence is fetched while the current one is walked -- its words and first exceptions, its
    // list length and the first NP batches of its list: what used to be three dependent global round trips at the head of
    // every sequence (list_n -> entries, word_off -> words, exc_off -> exceptions one by one) against ~50 LDS instructions
    // of work (the kernel ran at 43 % of its LDS-only ceiling, profiles/r04_c4_bench.json).
    // Two levels of loads hang on each other -- the sequence's offsets and counts (head), then its words, exceptions and list
    // entries (body) -- so the pipeline is two deep: the head of the sequence after next is requested while the body of the
    // next one goes out, and neither round trip is waited for at the head of a sequence.
    constexpr uint32_t NP = 4;
    struct Head { uint32_t seq, L, nnz; uint64_t woff, base, e0, e1; bool ok; };
    struct Ahead { RawSeq<M> raw; uint32_t nnz; uint64_t base; float r[NP]; uint32_t p[NP]; };
    auto head = [&](uint32_t t) {
        Head h;
        h.seq = pick_sequence(a.sv, t);
        h.ok = !(a.sv.mask && !a.sv.mask[h.seq]);
        h.L = a.sv.len[h.seq];
        h.nnz = a.list_n[h.seq];
        h.woff = a.sv.word_off[h.seq];
        h.base = a.sv.pos_off[h.seq];
        h.e0 = a.sv.exc_off[h.seq];
        h.e1 = a.sv.exc_off[h.seq + 1];
        return h;
    };
    auto body = [&](const Head& hd) {
        Ahead h;
        h.raw.seq = hd.seq; h.raw.ok = hd.ok; h.raw.L = hd.L; h.raw.e0 = hd.e0; h.raw.e1 = hd.e1;
        h.nnz = hd.nnz; h.base = hd.base;
        const uint32_t* wp = a.sv.words + hd.woff;
        const uint32_t nw = (hd.L + 15u) >> 4;
        const uint32_t wi0 = ((uint32_t)lane * M) >> 4;
        h.raw.w[0] = (wi0 >= 1u && wi0 - 1u < nw) ? wp[wi0 - 1u] : 0u;
#pragma unroll
        for (int i = 0; i < RawSeq<M>::NSEL; i++) h.raw.w[i + 1] = (wi0 + i < nw) ? wp[wi0 + i] : 0u;
#pragma unroll
        for (int i = 0; i < kRawExc; i++) h.raw.ex[i] = (hd.e0 + (uint64_t)i < hd.e1) ? a.sv.exc[hd.e0 + (uint64_t)i] : make_uint2(0xffffffffu, 0u);
#pragma unroll
        for (uint32_t u = 0; u < NP; u++) {
            const uint32_t idx = (uint32_t)lane + u * 64u;
            const bool ok = idx < hd.nnz;
            h.r[u] = ok ? a.list_r[hd.base + idx] : 0.0f;
            h.p[u] = ok ? (uint32_t)a.list_p[hd.base + idx] : 0u;
        }
        return h;
    };
    uint32_t t = blockIdx.x * waves_per_block + wave;
    Ahead nxt{};
    Head h1{};
    if (t < a.sv.count) nxt = body(head(t));
    if (t + total_waves < a.sv.count) h1 = head(t + total_waves);
    for (; t < a.sv.count; t += total_waves) {
        const Ahead cur = nxt;
        if (t + total_waves < a.sv.count) nxt = body(h1);                      // its head was requested an iteration ago
        if (t + 2u * total_waves < a.sv.count) h1 = head(t + 2u * total_waves);
        if (!cur.raw.ok) continue;
        const uint32_t nnz = cur.nnz;
        if (nnz == 0u) continue;
        const uint32_t L = cur.raw.L;
        const uint32_t LW1 = L - W + 1u;
        const uint32_t p0 = (uint32_t)lane * M;
        const uint64_t base = cur.base;
        float r_nxt[NP];
        uint32_t p_nxt[NP];
#pragma unroll
        for (uint32_t u = 0; u < NP; u++) { r_nxt[u] = cur.r[u]; p_nxt[u] = cur.p[u]; }
        auto request = [&](uint32_t e) {                      // lists beyond the first NP batches: the next NP while these are walked
#pragma unroll
            for (uint32_t u = 0; u < NP; u++) {
                const uint32_t idx = e + u * 64u;
                const bool ok = idx < nnz;
                r_nxt[u] = ok ? a.list_r[base + idx] : 0.0f;
                p_nxt[u] = ok ? (uint32_t)a.list_p[base + idx] : 0u;
            }
        };
        uint32_t y[M];
        decode_raw<M>(cur.raw, a.sv, Y, LW1, lane, y);               // EM.cpp:236: positions >= LW1 take no part (row Y)
#pragma unroll
        for (int m = 0; m < M; m++) ybuf[p0 + m] = (unsigned short)y[m];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (uint32_t e = (uint32_t)lane; e < nnz; e += 64u * NP) {
            float r[NP];
            uint32_t p[NP];
#pragma unroll
            for (uint32_t u = 0; u < NP; u++) { r[u] = r_nxt[u]; p[u] = p_nxt[u]; }
            if (e - (uint32_t)lane + 64u * NP < nnz) request(e + 64u * NP);      // wave-uniform
#pragma unroll
            for (uint32_t u = 0; u < NP; u++) {
                if (e + u * 64u >= nnz) continue;
                const unsigned long long Fe = to_fixed40(r[u] * a.fix_scale);
                uint32_t q = p[u] + 1u - W + j0;                          // position of the window's column j0
                unsigned long long* ncol = n_lds + copy;
                // eight columns' y first, then their adds: one LDS round trip per batch (row Y is a dump: no check)
                for (uint32_t jb = j0; jb < j1; jb += 8u, q += 8u, ncol += 8u * cstride) {
                    uint32_t yy[8];
#pragma unroll
                    for (int c = 0; c < 8; c++) yy[c] = (jb + c < j1) ? (uint32_t)ybuf[q + c] : Y;
#pragma unroll
                    for (int c = 0; c < 8; c++)
                        if (jb + c < j1) atomicAdd(&ncol[(size_t)c * cstride + (yy[c] << logC)], Fe);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();                              // ybuf is rewritten for the next sequence
    }
    lds_drain();
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < nc * Y; i += blockDim.x) {          // i = y*nc + j: runs of nc consecutive cells
        const uint32_t yy = i / nc, j = i - yy * nc;
        unsigned long long acc = 0ull;
        for (uint32_t c = 0; c < (1u << logC); c++) acc += n_lds[(((size_t)j * Ys + yy) << logC) + c];
        if (acc) acc_add(a.acc + (size_t)yy * W + j0 + j, (long long)acc);
    }
}

// ---- log-odds scorer ----------------------------------------------------------------------
template <int M, int THREADS>
__global__ void __launch_bounds__(THREADS) k_score(ScoreKernelArgs a) {
    extern __shared__ float lds[];
    const uint32_t W = a.W, Y = a.Y, Ys = a.Y + 1u;
    float* s_lds = lds;                                  // [W][Y+1], row Y = 0.0f
    for (uint32_t i = threadIdx.x; i < W * Ys; i += blockDim.x) s_lds[i] = a.s[i];
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t waves_per_block = blockDim.x >> 6;
    const uint32_t total_waves = gridDim.x * waves_per_block;

    for (uint32_t t = blockIdx.x * waves_per_block + wave; t < a.sv.count; t += total_waves) {
        const uint32_t seq = pick_sequence(a.sv, t);
        if (a.sv.mask && !a.sv.mask[seq]) continue;
        const uint32_t L = a.sv.len[seq];
        const uint32_t p0 = (uint32_t)lane * M;
        uint32_t y[M];
        decode_positions<M>(a.sv, seq, L, Y, L, lane, y);   // full windows (ScoreSeqSet.cpp:49-54)

        float U[M];
        const float* sj = s_lds;
#pragma unroll
        for (int m = 0; m < M; m++) U[m] = sj[y[m]];        // 0.0f + s == s
        for (uint32_t j = 1; j < W; j++) {
            sj += Ys;
            const float carry = wave_shr1(0.0f, U[M - 1]);
#pragma unroll
            for (int m = M - 1; m >= 1; m--) U[m] = U[m - 1] + sj[y[m]];
            U[0] = carry + sj[y[0]];
        }
        float best = -FLT_MAX;                              // ScoreSeqSet.cpp:46
        uint32_t best_i = 0;
        float* mo = a.mops ? a.mops + a.mops_off[seq] : nullptr;
#pragma unroll
        for (int m = 0; m < M; m++) {
            const uint32_t p = p0 + m;
            if (p + 1u >= W && p < L) {
                const uint32_t i = p + 1u - W;
                if (mo) mo[i] = U[m];
                if (U[m] > best) { best = U[m]; best_i = i; }
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {                  // first arg-max over the wave
            const float ob = __shfl_xor(best, o, 64);
            const uint32_t oi = __shfl_xor(best_i, o, 64);
            const bool take = (ob > best) || (ob == best && oi < best_i);
            best = take ? ob : best;
            best_i = take ? oi : best_i;
        }
        if (lane == 0) { a.zoops[seq] = best; a.z[seq] = best_i; }
    }
}


template <int M, int THREADS>
int launch_em_variant(bool accum, bool write_r, const EmKernelArgs& a, uint32_t blocks, uint32_t threads,
                      size_t lds, hipStream_t st) {
    int rc;
    if (write_r) {
        if ((rc = allow_lds(reinterpret_cast<const void*>(&k_em_seq<M, false, true, THREADS>), lds))) return rc;
        if (blocks == kPrimeOnly) return prime_kernel(reinterpret_cast<const void*>(&k_em_seq<M, false, true, THREADS>));
        hipLaunchKernelGGL((k_em_seq<M, false, true, THREADS>), dim3(blocks), dim3(threads), lds, st, a);
    } else if (accum) {
        if ((rc = allow_lds(reinterpret_cast<const void*>(&k_em_seq<M, true, false, THREADS>), lds))) return rc;
        if (blocks == kPrimeOnly) return prime_kernel(reinterpret_cast<const void*>(&k_em_seq<M, true, false, THREADS>));
        hipLaunchKernelGGL((k_em_seq<M, true, false, THREADS>), dim3(blocks), dim3(threads), lds, st, a);
    } else {
        if ((rc = allow_lds(reinterpret_cast<const void*>(&k_em_seq<M, false, false, THREADS>), lds))) return rc;
        if (blocks == kPrimeOnly) return prime_kernel(reinterpret_cast<const void*>(&k_em_seq<M, false, false, THREADS>));
        hipLaunchKernelGGL((k_em_seq<M, false, false, THREADS>), dim3(blocks), dim3(threads), lds, st, a);
    }
    return BAMM_OK;
}

}  // namespace

size_t em_lds_bytes(uint32_t W, uint32_t Y, bool accum, uint32_t logC, size_t scratch) {
    size_t floats = (size_t)((W + 3) / 4) * 4 * (Y + 1) + (accum ? (2 * (size_t)W * (Y + 1)) << logC : 0);
    return floats * sizeof(float) + 16 * 3 * sizeof(double) + (accum ? scratch : 0);
}

// per-wave LDS scratch of the sparse M-step: [cap x u64 addend][cap x u16 slot][64*M x u16 y]
uint32_t sparse_cap_for(int M) { return M <= 16 ? (uint32_t)std::min(192, 64 * M) : 0u; }
size_t sparse_wave_bytes(int M) {
    const size_t cap = sparse_cap_for(M);
    return cap ? ((cap * 8 + cap * 2 + (size_t)64 * M * 2 + 7) & ~size_t(7)) : 0;
}

// largest number of private count-table copies (power of two <= 16) that still lets
// `blocks_per_cu` blocks share the 160 KiB of a CU
uint32_t pick_log_copies(uint32_t W, uint32_t Y, uint32_t blocks_per_cu, size_t scratch) {
    const size_t budget = (160 * 1024) / (blocks_per_cu ? blocks_per_cu : 1);
    uint32_t logC = 0;
    while (logC < 4 && em_lds_bytes(W, Y, true, logC + 1, scratch) <= budget) logC++;
    return logC;
}

#define BAMM_FOR_EACH_MCLASS(X) \
    X(0, 1, 1024) X(1, 2, 1024) X(2, 3, 1024) X(3, 4, 1024) X(4, 5, 1024) X(5, 6, 1024) X(6, 7, 1024) \
    X(7, 8, 1024) X(8, 10, 768) X(9, 12, 768) X(10, 14, 768) X(11, 16, 768) X(12, 20, 768)             \
    X(13, 24, 512) X(14, 28, 512) X(15, 32, 512) X(16, 40, 512) X(17, 48, 512) X(18, 56, 512) X(19, 64, 512) \
    X(20, 80, 256) X(21, 96, 256) X(22, 128, 256)

int launch_em_seq(int mclass, bool accum, bool write_r, const EmKernelArgs& a, uint32_t blocks,
                  uint32_t threads, hipStream_t st) {
    const size_t lds = em_lds_bytes(a.W, a.Y, accum, a.logC, (size_t)a.sparse_wave_bytes * (threads / 64u));
    if (lds > 160 * 1024) {
        set_error("odds/count tables need %zu bytes of LDS (> 160 KiB): K=%u W=%u is outside the fused kernel's envelope",
                  lds, a.K, a.W);
        return BAMM_ERR_UNSUPPORTED;
    }
    if (threads > max_threads_for_mclass(mclass) || (threads & 63u) || threads == 0 || blocks == 0) {
        set_error("bad launch geometry %u x %u for M class %d", blocks, threads, mclass);
        return BAMM_ERR_ARG;
    }
    switch (mclass) {
#define X(idx, M, T)                                                   \
    case idx:                                                          \
        if (int rc = launch_em_variant<M, T>(accum, write_r, a, blocks, threads, lds, st)) return rc; \
        break;
        BAMM_FOR_EACH_MCLASS(X)
#undef X
        default:
            set_error("no kernel for M class %d", mclass);
            return BAMM_ERR_UNSUPPORTED;
    }
    BAMM_HIP(hipGetLastError());
    return BAMM_OK;
}


size_t e_slice_lds_bytes(uint32_t cols, uint32_t Y) {
    return ((((size_t)cols * (Y + 1)) + 1) & ~size_t(1)) * sizeof(float) + 16 * 3 * sizeof(double);
}
size_t m_slice_lds_bytes(uint32_t cols, uint32_t Y, uint32_t logC) { return (((size_t)cols * (Y + 1)) << logC) * 8; }
// per-wave scratch of the sparse M-slice: [cap x u64 list][64*M x u16 y]
size_t m_slice_wave_bytes(int M, uint32_t cap) { return cap ? (((size_t)cap * 8 + (size_t)64 * M * 2 + 15) & ~size_t(15)) : 0; }

int launch_e_slice(int mclass, const EmKernelArgs& a, uint32_t j0, uint32_t j1, bool last, uint32_t blocks,
                   uint32_t threads, hipStream_t st) {
    const size_t lds = e_slice_lds_bytes(j1 - j0, a.Y);
    if (lds > 160 * 1024 || j1 <= j0) { set_error("bad E slice [%u,%u)", j0, j1); return BAMM_ERR_UNSUPPORTED; }
    switch (mclass) {
#define X(idx, M, T)                                                                                   \
    case idx:                                                                                          \
        if (int rc = allow_lds(reinterpret_cast<const void*>(&k_e_slice<M, T>), lds)) return rc;                 \
        hipLaunchKernelGGL((k_e_slice<M, T>), dim3(blocks), dim3(threads), lds, st, a, j0, j1, last ? 1 : 0); \
        break;
        BAMM_FOR_EACH_MCLASS(X)
#undef X
        default: set_error("no kernel for M class %d", mclass); return BAMM_ERR_UNSUPPORTED;
    }
    BAMM_HIP(hipGetLastError());
    return BAMM_OK;
}

int launch_m_slice(int mclass, const EmKernelArgs& a, uint32_t j0, uint32_t j1, bool r_reversed, uint32_t blocks,
                   uint32_t threads, hipStream_t st) {
    const size_t lds = m_slice_lds_bytes(j1 - j0, a.Y, a.logC) + (size_t)a.sparse_wave_bytes * (threads / 64u);
    if (lds > 160 * 1024 || j1 <= j0) { set_error("bad M slice [%u,%u)", j0, j1); return BAMM_ERR_UNSUPPORTED; }
    switch (mclass) {
#define X(idx, M, T)                                                                                   \
    case idx:                                                                                          \
        if (int rc = allow_lds(reinterpret_cast<const void*>(&k_m_slice<M, T>), lds)) return rc;                 \
        hipLaunchKernelGGL((k_m_slice<M, T>), dim3(blocks), dim3(threads), lds, st, a, j0, j1, r_reversed ? 1 : 0); \
        break;
        BAMM_FOR_EACH_MCLASS(X)
#undef X
        default: set_error("no kernel for M class %d", mclass); return BAMM_ERR_UNSUPPORTED;
    }
    BAMM_HIP(hipGetLastError());
    return BAMM_OK;
}

size_t m_list_lds_bytes(uint32_t cols, uint32_t Y, uint32_t logC, int M, uint32_t waves) {
    return m_slice_lds_bytes(cols, Y, logC) + (size_t)waves * (64u * (size_t)M + 8u) * 2u;
}

int launch_m_list(int mclass, const EmKernelArgs& a, uint32_t j0, uint32_t j1, uint32_t blocks, uint32_t threads, hipStream_t st) {
    // 16 positions per lane (config 4): the walk needs far fewer registers than the E pass the class's block size was chosen
    // for -- 16 waves per block where their copies of the decoded sequence still fit beside the count slice
    if (kMClasses[mclass] == 16 && threads == 768u && j1 > j0 && m_list_lds_bytes(j1 - j0, a.Y, a.logC, 16, 16u) <= 160 * 1024) {
        const size_t lds16 = m_list_lds_bytes(j1 - j0, a.Y, a.logC, 16, 16u);
        if (int rc = allow_lds(reinterpret_cast<const void*>(&k_m_list<16, 1024>), lds16)) return rc;
        hipLaunchKernelGGL((k_m_list<16, 1024>), dim3(blocks), dim3(1024), lds16, st, a, j0, j1);
        BAMM_HIP(hipGetLastError());
        return BAMM_OK;
    }
    const size_t lds = m_list_lds_bytes(j1 - j0, a.Y, a.logC, kMClasses[mclass], threads / 64u);
    if (lds > 160 * 1024 || j1 <= j0) { set_error("bad list M slice [%u,%u)", j0, j1); return BAMM_ERR_UNSUPPORTED; }
    switch (mclass) {
#define X(idx, M, T)                                                                                   \
    case idx:                                                                                          \
        if (int rc = allow_lds(reinterpret_cast<const void*>(&k_m_list<M, T>), lds)) return rc;        \
        hipLaunchKernelGGL((k_m_list<M, T>), dim3(blocks), dim3(threads), lds, st, a, j0, j1);         \
        break;
        BAMM_FOR_EACH_MCLASS(X)
#undef X
        default: set_error("no kernel for M class %d", mclass); return BAMM_ERR_UNSUPPORTED;
    }
    BAMM_HIP(hipGetLastError());
    return BAMM_OK;
}

int launch_score(int mclass, const ScoreKernelArgs& a, uint32_t blocks, uint32_t threads, hipStream_t st) {
    const size_t lds = (size_t)a.W * (a.Y + 1) * sizeof(float);
    if (lds > 160 * 1024) {
        set_error("log-odds table needs %zu bytes of LDS (> 160 KiB)", lds);
        return BAMM_ERR_UNSUPPORTED;
    }
    if (threads > max_threads_for_mclass(mclass) || (threads & 63u) || threads == 0 || blocks == 0) {
        set_error("bad launch geometry %u x %u for M class %d", blocks, threads, mclass);
        return BAMM_ERR_ARG;
    }
    switch (mclass) {
#define X(idx, M, T)                                                                                   \
    case idx:                                                                                          \
        if (int rc = allow_lds(reinterpret_cast<const void*>(&k_score<M, T>), lds)) return rc;                 \
        hipLaunchKernelGGL((k_score<M, T>), dim3(blocks), dim3(threads), lds, st, a);                  \
        break;
        BAMM_FOR_EACH_MCLASS(X)
#undef X
        default:
            set_error("no kernel for M class %d", mclass);
            return BAMM_ERR_UNSUPPORTED;
    }
    BAMM_HIP(hipGetLastError());
    return BAMM_OK;
}


}  // namespace bamm
