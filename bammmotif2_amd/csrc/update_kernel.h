// The model update of one EM pass with every order of n and v staged in LDS, as a device function:
//   EM.cpp:247-254 (marginalise), Motif::updateV init/Motif.h:95-136, EM::optimize_q EM.cpp:515,
//   v_diff EM.cpp:102-108, the stop rule EM.cpp:117-118, Motif::calculateLinearS init/Motif.cpp:485-494
// (file:line relative to /root/reference/src).
//
// Two callers run the SAME code (same float formulas in the same order: bit-identical models):
//   * k_update<true> (kernels.hip): one block after the pass's all-reduce, the accumulator consumed and zeroed;
//   * the block prologue of the NEXT pass's first sequence kernel (k_em_mix / k_em_grp, "fused update"): every
//     block recomputes the <= 2048-cell update from the all-reduced accumulator of the previous pass into its own
//     LDS -- 10 KB read through L2 and a few hundred cycles of LDS arithmetic instead of a kernel launch between
//     two passes -- and block 0 of the launch (the "writer") alone publishes n, v, s, q, status and trace.  An
//     iteration is then ONE kernel + ONE collective.  The accumulator is a ring of three: pass p adds into slot
//     p mod 3, the next kernel's blocks read it, its writer clears the slot pass p + 2 will add into (every reader
//     of that slot finished with the kernel before: stream order).
#pragma once
#include "device_utils.h"
#include "phase_clock.h"

namespace bamm {
namespace {

struct UpdateOut {
    float q;        // q of the next pass (every thread of the block)
    bool fired;     // the stop rule fired (every thread; false when a.stop == nullptr)
};

// ---- the pass's all-reduce in the TAIL of the sequence kernel (common.h: PeerArgs) ----
// An inbox entry carries one int64 word of a peer's totals as two self-validating 8-byte halves {32 bits of data, the
// pass's 32-bit sequence number}: an aligned 8-byte store is single-copy atomic, so a reader that finds the sequence
// number in a half has that half's data -- no fence between data and flag, on either side (RCCL's LL protocol).
struct alignas(16) PeerEntry { unsigned long long lo, hi; };

typedef const __attribute__((address_space(4))) PeerArgs* PeerArgsK;   // in the kernel-argument segment: scalar loads where used

__device__ __forceinline__ PeerEntry* peer_buffer(void* inbox, uint32_t world, uint32_t stride, uint32_t slot, uint32_t src) {
    return reinterpret_cast<PeerEntry*>(inbox) + ((size_t)slot * world + src) * stride;
}

// Called by EVERY thread of EVERY block at the end of the pass's (one) sequence launch, after the block's atomics into
// the accumulator were issued; `lds`: 16 bytes of LDS nobody else uses any more.
//   1. a block waits until its own atomics have been performed (they run at the device's coherence point) and draws a
//      ticket; the block that draws the last one knows the accumulator holds this GPU's totals;
//   2. it reads them back (device-scope loads), stores them into its buffer of every peer's inbox as self-validating
//      entries (system-scope stores over xGMI) ...
//   3. ... and collects the peers' entries from its own inbox as they arrive, every poll bounded by the wall-clock
//      deadline: totals + peers -> the accumulator, in place.  The kernel boundary hands the all-reduced accumulator to
//      whatever follows on the stream (the next pass's fused update, k_update), as RCCL's in-place all-reduce would.
// A deadline that passes raises p.err: this launch leaves the accumulator as it is, every later launch of the handle
// returns at entry, and the host reports BAMM_ERR_COMM at its next synchronisation.
// (the arguments stay where they are, in the kernel-argument segment: a copy in registers across the sequence loop costs the
// loop spilled SGPRs, a local copy of the struct costs the kernel scratch memory for the peer[] array)
__device__ __forceinline__ void peer_allreduce_tail(PeerArgsK pk, long long* acc, uint32_t* lds) {
    struct { uint32_t world, rank, stride, words, slot; unsigned long long seq, timeout_ticks; void* inbox; uint32_t* ticket; uint32_t* err; } p;
    p.world = pk->world; p.rank = pk->rank; p.stride = pk->stride; p.words = pk->words; p.slot = pk->slot; p.seq = pk->seq;
    p.timeout_ticks = pk->timeout_ticks; p.inbox = pk->inbox; p.ticket = pk->ticket; p.err = pk->err;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this thread's accumulator atomics are performed
    __syncthreads();
    if (threadIdx.x == 0) {
        // release: the block's accumulator atomics (performed: vmcnt above) are ordered before the ticket for every other
        // agent-scope observer; acquire: the block that draws the last ticket sees all of them when it reads the totals back.
        // (One fence per block at the kernel's end; the vmcnt + a relaxed ticket sufficed on one device in every run, but
        // that leans on gfx9 behaviour the memory model does not promise across XCDs.)
        const uint32_t t = __hip_atomic_fetch_add(p.ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        lds[0] = (t + 1u == gridDim.x) ? 1u : 0u;
        lds[1] = 0u;                                         // a lane's deadline passed
    }
    __syncthreads();
    if (lds[0] == 0u) return;
    const unsigned long long tag = (unsigned long long)(uint32_t)p.seq << 32;
    const unsigned long long t0 = wall_clock64();
    // (peer[] is indexed by compile-time constants only: a run-time index into a kernel-argument array makes the
    // compiler copy it to scratch memory, and a kernel with scratch pays for it on every launch)
    for (uint32_t i = threadIdx.x; i < p.words; i += blockDim.x) {          // everything out first: the peers are waiting for it
        const unsigned long long x = (unsigned long long)__hip_atomic_load(acc + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long lo = tag | (x & 0xffffffffull), hi = tag | (x >> 32);
#pragma unroll
        for (uint32_t d = 0; d < kPeerMaxWorld; d++)
            if (d < p.world && d != p.rank) {
                PeerEntry* e = peer_buffer(pk->peer[d], p.world, p.stride, p.slot, p.rank) + i;
                __hip_atomic_store(&e->lo, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(&e->hi, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
    }
    for (uint32_t i = threadIdx.x; i < p.words; i += blockDim.x) {          // ... then the peers' words, as they arrive
        unsigned long long sum = (unsigned long long)__hip_atomic_load(acc + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bool late = false;
        for (uint32_t r = 0; r < p.world && !late; r++) {
            if (r == p.rank) continue;
            const PeerEntry* e = peer_buffer(p.inbox, p.world, p.stride, p.slot, r) + i;
            unsigned long long a, b;
            for (;;) {
                a = __hip_atomic_load(&e->lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                b = __hip_atomic_load(&e->hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                if ((a >> 32) == (tag >> 32) && (b >> 32) == (tag >> 32)) break;
                if (wall_clock64() - t0 > p.timeout_ticks) { late = true; lds[1] = 1u + r; break; }
                __builtin_amdgcn_s_sleep(2);
            }
            sum += (a & 0xffffffffull) | (b << 32);
        }
        // every word is its own lane's: nobody reads acc[i] again in this launch
        if (!late) __hip_atomic_store(acc + i, (long long)sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (lds[1] != 0u) __hip_atomic_store(p.err, lds[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(p.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);    // the next launch follows in stream order
    }
}

// sum of the 4^D leaves under `row` in the reference's nesting (EM.cpp:247-254): each level adds its four children in
// ascending order, from 0.0f; the loads of a cell are independent and issued together
template <int D>
__device__ __forceinline__ float update_count_tree(const float* nK, uint32_t W, uint32_t j, uint32_t row, uint32_t stride) {
    if constexpr (D == 0) return nK[(size_t)row * W + j];
    else {
        float s = 0.0f;
#pragma unroll
        for (uint32_t d = 0; d < 4; d++) s += update_count_tree<D - 1>(nK, W, j, row + d * stride, stride * 4u);
        return s;
    }
}

// CONSUME: the standalone kernel's semantics -- the accumulator is zeroed as it is read.  Fused callers leave it
//          (other blocks are still reading) and the writer clears a.acc_zero instead.
// s_lds:   nullable; receives the next pass's odds table [W][Y+1] INSTEAD of a.s: the caller's prologue builds its grouped
//          tables from it and, after its next barrier, copies it out coalesced (its block's global copy for the fix
//          lanes; the writer block: a.s) -- publish_odds() below
// Every thread of the block must call this (it synchronises the block).
//
// Two barriers instead of one per order: the reference's chain n_K -> n_(K-1) -> ... -> n_0, v_0 -> v_1 -> ... -> v_K
// is recomputed per cell from what the previous phase left in LDS -- the SAME float expressions on the same operands
// in the same order (sums of four ascending, one division per order), so every value has the bits the phased form
// gave it; only the waiting is gone (7 block barriers and 4 exposed global-load latencies cost 4.3 us per launch).
//   phase A  global -> LDS: n_K from the integer accumulator, A, vbg (every load of the update issued up front)
//   phase B  n_k for k < K, each cell summed straight from n_K (EM.cpp:247-254: four rows of order k+1 per cell, ascending)
//   phase C  per cell of every order: the v chain from order 0 up (Motif.h:100-135), v_diff, the odds table
template <bool CONSUME>
__device__ __forceinline__ UpdateOut model_update_lds(const UpdateArgs& a, unsigned char* upd, float* s_lds, bool writer) {
    const uint32_t K = a.K, W = a.W;
    const uint32_t YK = 1u << (2 * (K + 1));
    const uint32_t tid = threadIdx.x, nt = blockDim.x;
    auto voff = [W](uint32_t k) { return (size_t)W * (((size_t(1) << (2 * (k + 1))) - 4) / 3); };
    const size_t vsz = voff(K + 1);
    const uint32_t nA = (K + 1u) * W, nB = ((1u << (2 * (a.Kbg + 2))) - 4u) / 3u;   // A[k][j]; vbg orders 0..Kbg
    float* const n = reinterpret_cast<float*>(upd);                  // all orders, flat [k][y][j]
    float* const Al = n + vsz;                                       // [(K+1) * W]
    float* const bl = Al + ((nA + 1u) & ~1u);                        // vbg, orders 0..Kbg
    double* const shd = reinterpret_cast<double*>(n + ((vsz + 1) & ~size_t(1)) + ((nA + 1u) & ~1u) + ((nB + 1u) & ~1u));   // [16] v_diff partials, [3] statistics
    double* const stat3 = shd + 16;                                  // llh, sum_r, n_seqs, [3] = non-finite flag
    const float* const v_old = a.v_old ? a.v_old : a.v;             // the model the pass ran with (v_diff)
    long long* const acc = a.acc;
    const bool want_diff = writer || a.stop != nullptr;              // block-uniform

    // ---- phase A
    // the old model's top order for v_diff: at most two cells per thread (update_fits_lds), requested with everything
    // else instead of inside the chains (a global round trip in phase C: 1 us)
    float vo0 = 0.0f, vo1 = 0.0f;
    if (want_diff) {
        if (tid < YK * W) vo0 = v_old[voff(K) + tid];
        if (tid + nt < YK * W) vo1 = v_old[voff(K) + tid + nt];
    }
    if (tid == 0) stat3[3] = 0.0;
    const float q_in = *a.q;                                         // issued with the other loads of the update, used last
    const float llh_before = (a.stop != nullptr && a.llh_prev_from_status) ? *a.llh_in : a.llh_prev;
    float* nK = n + voff(K);
    for (uint32_t i = tid; i < YK * W; i += nt) {
        nK[i] = (float)((double)acc[i] * a.count_unit);
        if (CONSUME) acc[i] = 0ll;
    }
    for (uint32_t i = tid; i < nA; i += nt) Al[i] = a.A[i];
    for (uint32_t i = tid; i < nB; i += nt) bl[i] = a.vbg[i];
    if (tid < 3) {
        const long long x = acc[(size_t)YK * W + tid];
        if (CONSUME) acc[(size_t)YK * W + tid] = 0ll;
        stat3[tid] = tid == 0 ? (double)x / kLlhScale : (tid == 1 ? (double)x / kSumrScale : stat_nseq(x));
        if (tid == 2 && stat_bad(x)) stat3[3] = 1.0;      // some block's statistics were not finite
    }
    if (writer && a.acc_zero != nullptr)                             // the ring slot two passes ahead
        for (uint32_t i = tid; i < YK * W + 3u; i += nt) a.acc_zero[i] = 0ll;
    BAMM_PHASE(8);
    __syncthreads();
    BAMM_PHASE(9);
    // ---- phase B: n[k][y][j] = (((0 + c_0) + c_1) + c_2) + c_3 over the four rows c_d = n[k+1][d * 4^(k+1) + y][j] of the
    // next order, themselves sums of four, down to n_K: nested loops over the K - k levels (at most four: the
    // tables fit LDS for K <= 4 only), each level summed from 0.0f upwards as the reference's += does
    // (all lower orders as ONE index space, a cell per thread and round: order after order the first threads walked
    // 16 + 4 dependent reads while the others waited -- 1.6 us of the fused update's 6, tools/phase_clock.py)
    for (uint32_t i = tid; i < (uint32_t)voff(K); i += nt) {
        uint32_t k = 0;
        while (i >= voff(k + 1)) k++;
        const uint32_t c = i - (uint32_t)voff(k), y = c / W, j = c % W;
        const uint32_t Yk = 1u << (2 * (k + 1));
        float r;
        if constexpr (!CONSUME) {                                    // fused into a sequence kernel: planned for K <= 2 only
            r = (K - k == 1u) ? update_count_tree<1>(nK, W, j, y, Yk) : update_count_tree<2>(nK, W, j, y, Yk);
        } else {
            switch (K - k) {
                case 1: r = update_count_tree<1>(nK, W, j, y, Yk); break;
                case 2: r = update_count_tree<2>(nK, W, j, y, Yk); break;
                case 3: r = update_count_tree<3>(nK, W, j, y, Yk); break;
                default: r = update_count_tree<4>(nK, W, j, y, Yk); break;
            }
        }
        n[i] = r;
    }
    BAMM_PHASE(10);
    __syncthreads();
    BAMM_PHASE(11);
    // ---- phase C: the v chain of a cell (y, j) of order k, from order 0 up (Motif.h:100-135)
    //   v[0][y0][j] = (n[0][y0][j] + A[0][j] * vbg[0][y0]) / (sum_y' n[0][y'][j] + A[0][j])
    //   v[kk][ykk][j] = j < kk ? v[kk-1][ykk mod 4^kk][j]
    //                          : (n[kk][ykk][j] + A[kk][j] * v[kk-1][ykk mod 4^kk][j]) / (n[kk-1][ykk / 4][j-1] + A[kk][j])
    auto v_cell = [&](uint32_t k, uint32_t y, uint32_t j) -> float {
        float sumN = 0.0f;
        for (uint32_t yy = 0; yy < 4; yy++) sumN += n[yy * W + j];
        const uint32_t y0 = y & 3u;
        float val = (n[y0 * W + j] + Al[j] * bl[y0]) / (sumN + Al[j]);
        for (uint32_t kk = 1; kk <= k; kk++) {
            if (j < kk) continue;                                    // the copy of the lower order's value
            const uint32_t ykk = y & ((1u << (2 * (kk + 1))) - 1u);
            const float* nkk = n + voff(kk);
            const float* nk1 = n + voff(kk - 1);
            const float Akj = Al[kk * W + j];
            val = (nkk[(size_t)ykk * W + j] + Akj * val) / (nk1[(size_t)(ykk >> 2) * W + j - 1u] + Akj);
        }
        return val;
    };
    double diff = 0.0;
    const uint32_t Ys = YK + 1u, Yb = 1u << (2 * (a.Kbg + 1));
    const float* const b = bl + (((size_t)Yb - 4) / 3);
    for (uint32_t k = 0; k <= K; k++) {
        const uint32_t Yk1 = 1u << (2 * (k + 1));
        if (k < K && !writer) continue;                              // the lower orders only leave the block through the writer
        for (uint32_t i = tid; i < Yk1 * W; i += nt) {
            const uint32_t y = i / W, j = i % W;
            const float nv = v_cell(k, y, j);
            if (k == K && want_diff)                                 // the old value was read in phase A: k_update updates v in place
                diff += (double)fabsf(nv - (i < nt ? vo0 : (i < 2u * nt ? vo1 : v_old[voff(K) + i])));
            if (writer) { a.v[voff(k) + i] = nv; a.n[voff(k) + i] = n[voff(k) + i]; }
            if (k == K) {
                const float sv = nv / b[y % Yb];                     // Motif.cpp:485-494
                if (s_lds != nullptr) s_lds[(size_t)j * Ys + y] = sv;
                else if (writer) a.s[(size_t)j * Ys + y] = sv;
            }
        }
    }
    for (uint32_t j = tid; j < W; j += nt) {                         // the neutral row
        if (s_lds != nullptr) s_lds[(size_t)j * Ys + YK] = 1.0f;
        else if (writer) a.s[(size_t)j * Ys + YK] = 1.0f;
    }
    BAMM_PHASE(12);
    // v_diff (EM.cpp:102-108): wave sums, then the wave results.  Only who needs it: the writer (status, trace)
    // and, inside optimize(), every block (the stop rule decides whether the block runs its pass)
    double v_diff = 0.0;
    if (want_diff) {
        double d = diff;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) d += __shfl_xor(d, o, 64);
        if ((tid & 63u) == 0u) shd[tid >> 6] = d;
        __syncthreads();
        for (uint32_t w = 0; w < (nt + 63u) / 64u; w++) v_diff += shd[w];      // same order in every thread
    }
    UpdateOut out;
    const double llh = stat3[3] != 0.0 ? (double)NAN : stat3[0], sum_r = stat3[1];
    const double nseq = a.n_seqs_override > 0.0 ? a.n_seqs_override : stat3[2];
    float q = q_in;
    if (a.optimize_q)                                                // EM.cpp:515; the host applies EM.cpp:99's `iteration <= 5`
        q = (float)((nseq - sum_r + 1.0) / (nseq + 2.0));
    out.q = q;
    out.fired = false;
    if (a.stop != nullptr) {                                         // EM.cpp:117-118
        out.fired = (float)v_diff < a.epsilon || ((float)llh - llh_before < 0 && a.opt_iteration > 10u);
    }
    if (writer && tid == 0) {
        const uint32_t it = *a.iteration + 1u;
        *a.iteration = it;
        *a.q_out = q;
        if (a.llh_out != nullptr) *a.llh_out = (float)llh;
        if (out.fired) *a.stop = 1u;
        a.status[0] = (float)llh;
        a.status[1] = (float)v_diff;
        a.status[2] = q;
        a.status[3] = (float)it;
        a.status[4] = (float)sum_r;
        a.status[5] = (float)nseq;
        if (a.status_mirror != nullptr) {
            // six self-validating 8-byte words {pass number | float bits}: optimize() polls them instead of waiting for an event on
            // the stream (4 us of stream time per pass); no fence between the words -- each carries its own tag (RCCL's LL idea)
            {
                const float f6[6] = {(float)llh, (float)v_diff, q, (float)it, (float)sum_r, (float)nseq};
                const unsigned long long tag = (unsigned long long)a.opt_iteration << 32;
#pragma unroll
                for (int i = 0; i < 6; i++) __hip_atomic_store(a.status_mirror + i, tag | (unsigned long long)__float_as_uint(f6[i]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
        if (a.trace && it - 1u < a.trace_cap) {
            a.trace[(size_t)(it - 1u) * 3 + 0] = (float)llh;
            a.trace[(size_t)(it - 1u) * 3 + 1] = (float)v_diff;
            a.trace[(size_t)(it - 1u) * 3 + 2] = q;
        }
    }
    return out;
}

// after the barrier that follows model_update_lds(..., s_lds, ...): the odds table leaves the block, coalesced
// (s_block nullable: a block that stops here -- the rule fired -- only publishes when it is the writer)
__device__ __forceinline__ void publish_odds(const UpdateArgs& a, const float* s_lds, float* s_block, bool writer) {
    const uint32_t cells = a.W * ((1u << (2 * (a.K + 1))) + 1u);
    for (uint32_t i = threadIdx.x; i < cells; i += blockDim.x) {
        const float x = s_lds[i];
        if (s_block != nullptr) s_block[i] = x;
        if (writer) a.s[i] = x;
    }
}

}  // namespace
}  // namespace bamm
