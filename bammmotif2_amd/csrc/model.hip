// gfx950 kernels that touch the MODEL, not the sequences: the odds table, the model update, the statistics read-out and the
// reduction of EM::mask's per-block partials.  A translation unit of their own (a code object of a few tens of KB): the HIP
// runtime loads a code object on the first launch of any kernel in it, and these are the first kernels every handle
// launches -- inside kernels.hip (4 MB of sequence-kernel instantiations) that first launch took 8-14 ms of every process
// whether or not it ever ran one of those (profiles/r05_first_call.txt).
//
// What they compute (reference lines, relative to /root/reference/src):
//   k_make_s        Motif::calculateLinearS init/Motif.cpp:485-494
//   k_update*       EM.cpp:247-254 (marginalise), Motif::updateV init/Motif.h:95-136, EM::optimize_q EM.cpp:515,
//                   v_diff EM.cpp:102-108, the stop rule EM.cpp:117-118
//   k_reduce_partials  the cross-thread reduction the reference gets from `omp parallel for reduction(+:llikelihood)`
//                   (EM.cpp:148) and the CAS float atomics (EM.cpp:203-215,240), for EM::mask's kernels
//   k_stat_only     EStep() alone: llh and the sum over r (EM.cpp:148,509-513)

#include "device_utils.h"
#include "update_kernel.h"

#include <algorithm>
#include <cfloat>

namespace bamm {

namespace {

// ---- EM::mask: sum the per-block partial tables of the masked kernels into the fused accumulator ------
// grid.x = ceil(W*Y/64) blocks of 64 cells x 16 groups, + one block for the three statistics
__global__ void __launch_bounds__(1024) k_reduce_partials(const unsigned long long* partial_n, const double* partial_stat,
                                                          uint32_t blocks, uint32_t W, uint32_t Y, long long* acc_out) {
    __shared__ unsigned long long sh[16][64];
    __shared__ double shd[16][4];
    const uint32_t C = partial_n ? W * Y : 0u;
    if (blockIdx.x + 1u == gridDim.x) {
        // statistics: llh, sum_r, n_seqs -- 256 groups of 4 lanes, summed over the lanes of a wave by
        // shuffles (a fixed tree: the same bits every run), then over the 16 waves
        const uint32_t k = threadIdx.x & 3u, grp = threadIdx.x >> 2;
        double t = 0.0;
        if (k < 3u)
            for (uint32_t b = grp; b < blocks; b += 256u) t += partial_stat[(size_t)b * 4 + k];
#pragma unroll
        for (int o = 4; o < 64; o <<= 1) t += __shfl_xor(t, o, 64);
        if ((threadIdx.x & 63u) < 4u) shd[threadIdx.x >> 6][threadIdx.x & 3u] = t;
        __syncthreads();
        if (threadIdx.x < 3u) {
            double acc = 0.0;
#pragma unroll
            for (int w = 0; w < 16; w++) acc += shd[w][threadIdx.x];
            acc_add_stat(acc_out, W * Y, threadIdx.x, acc);
        }
        return;
    }
    const uint32_t c = blockIdx.x * 64u + (threadIdx.x & 63u);
    const uint32_t g = threadIdx.x >> 6;
    unsigned long long acc = 0ull;
    if (c < C)
        for (uint32_t b = g; b < blocks; b += 16u) acc += partial_n[(size_t)b * C + c];
    sh[g][threadIdx.x & 63u] = acc;
    __syncthreads();
    if (g == 0 && c < C) {
        unsigned long long t = 0ull;
#pragma unroll
        for (int i = 0; i < 16; i++) t += sh[i][threadIdx.x];
        const uint32_t j = c / Y, y = c % Y;               // LDS layout [j][y] -> ABI layout [y][j]
        if (t) acc_add(acc_out + (size_t)y * W + j, (long long)t);
    }
}

// ---- s[j][y] = v[K][y][j] / vbg[Kbg][y mod 4^(Kbg+1)], pad row = 1  (Motif.cpp:485-494) -----
__global__ void k_make_s(const float* v, const float* vbg, uint32_t K, uint32_t W, uint32_t Kbg, float* s) {
    const uint32_t Y = 1u << (2 * (K + 1)), Ys = Y + 1u, Yb = 1u << (2 * (Kbg + 1));
    const float* vK = v + W * (((size_t)Y - 4) / 3);
    const float* b = vbg + (((size_t)Yb - 4) / 3);
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < W * Ys; i += gridDim.x * blockDim.x) {
        const uint32_t j = i / Ys, y = i % Ys;
        s[i] = (y == Y) ? 1.0f : vK[(size_t)y * W + j] / b[y % Yb];
    }
}

// ---- model update: one block ----------------------------------------------------------------
// IN_LDS: all orders of n and v are staged in LDS (8 bytes per cell: K <= 3 at usual widths).  The update is a
// chain of small dependent phases (marginalise order by order, then v order by order); run on the global
// arrays each phase pays a store -> barrier -> load round trip through the cache (7.3 us for 1680 cells), in LDS
// the chain costs a few hundred cycles and global memory sees one read of the accumulator and one write of the
// results.  Same formulas in the same order either way: bit-identical models.
template <bool IN_LDS>
__global__ void __launch_bounds__(1024) k_update(UpdateArgs a) {
    if (a.stop != nullptr && *a.stop != 0u) return;      // optimize(): the stop rule fired in an earlier pass

    extern __shared__ __align__(16) unsigned char upd_lds[];
    if constexpr (IN_LDS) {
        // the same device function the sequence kernels run in their prologue when the update is fused into the
        // next pass (update_kernel.h): here one block, the accumulator consumed and zeroed
        (void)model_update_lds<true>(a, upd_lds, nullptr, true);
    } else {
    __shared__ double shd[16];
    __shared__ double stat3[4];                            // llh, sum_r, n_seqs, non-finite flag
    const uint32_t K = a.K, W = a.W;
    const uint32_t YK = 1u << (2 * (K + 1));
    const uint32_t tid = threadIdx.x, nt = blockDim.x;
    auto voff = [W](uint32_t k) { return (size_t)W * (((size_t(1) << (2 * (k + 1))) - 4) / 3); };
    float* const n = a.n;                                  // all orders, flat [k][y][j]
    float* const v = a.v;

    // order-K counts from the (all-reduced) integer accumulator, which is left zeroed for the next pass
    float* nK = n + voff(K);
    if (tid == 3) stat3[3] = 0.0;
    for (uint32_t i = tid; i < YK * W; i += nt) {
        nK[i] = (float)((double)a.acc[i] * a.count_unit);
        a.acc[i] = 0ll;
    }
    if (tid < 3) {
        const long long x = a.acc[(size_t)YK * W + tid];
        a.acc[(size_t)YK * W + tid] = 0ll;
        stat3[tid] = tid == 0 ? (double)x / kLlhScale : (tid == 1 ? (double)x / kSumrScale : stat_nseq(x));
        if (tid == 2 && stat_bad(x)) stat3[3] = 1.0;      // some block's statistics were not finite
    }
    if (a.acc_zero != nullptr)
        for (uint32_t i = tid; i < YK * W + 3u; i += nt) a.acc_zero[i] = 0ll;
    __syncthreads();
    // EM.cpp:247-254: n[k-1][y mod 4^k][j] += n[k][y][j], y ascending (same float order)
    for (uint32_t k = K; k > 0; k--) {
        const float* nk = n + voff(k);
        float* nk1 = n + voff(k - 1);
        const uint32_t Yk = 1u << (2 * k);                 // rows of order k-1
        for (uint32_t i = tid; i < Yk * W; i += nt) {
            const uint32_t y2 = i / W, j = i % W;
            float acc = 0.0f;
#pragma unroll
            for (uint32_t bse = 0; bse < 4; bse++) acc += nk[(size_t)(bse * Yk + y2) * W + j];
            nk1[i] = acc;
        }
        __syncthreads();
    }
    // Motif.h:100-118: order 0
    double diff = 0.0;
    for (uint32_t j = tid; j < W; j += nt) {
        float sumN = 0.0f;
        for (uint32_t y = 0; y < 4; y++) sumN += n[y * W + j];
        for (uint32_t y = 0; y < 4; y++) {
            const float nv = (n[y * W + j] + a.A[j] * a.vbg[y]) / (sumN + a.A[j]);
            if (K == 0) diff += (double)fabsf(nv - a.v[y * W + j]);
            v[y * W + j] = nv;
        }
    }
    __syncthreads();
    // Motif.h:121-135: orders 1..K
    for (uint32_t k = 1; k <= K; k++) {
        const float* nk = n + voff(k);
        const float* nk1 = n + voff(k - 1);
        float* vk = v + voff(k);
        const float* vk1 = v + voff(k - 1);
        const float* Ak = a.A + (size_t)k * W;
        const uint32_t Yk1 = 1u << (2 * (k + 1)), Yk = 1u << (2 * k);
        for (uint32_t i = tid; i < Yk1 * W; i += nt) {
            const uint32_t y = i / W, j = i % W;
            const uint32_t y2 = y % Yk, yk = y / 4;
            float nv;
            if (j < k) nv = vk1[(size_t)y2 * W + j];
            else nv = (nk[i] + Ak[j] * vk1[(size_t)y2 * W + j]) / (nk1[(size_t)yk * W + j - 1] + Ak[j]);
            if (k == K) diff += (double)fabsf(nv - a.v[voff(K) + i]);
            vk[i] = nv;
        }
        __syncthreads();
    }
    // v_diff (EM.cpp:102-108): wave sums, then the 16 wave results
    {
        double d = diff;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) d += __shfl_xor(d, o, 64);
        if ((tid & 63u) == 0u) shd[tid >> 6] = d;
        __syncthreads();
        if (tid == 0) {
            double t = 0.0;
            for (uint32_t w = 0; w < (nt + 63u) / 64u; w++) t += shd[w];
            shd[0] = t;
        }
        __syncthreads();
    }
    const double v_diff = shd[0];
    // next E-step's odds table (Motif.cpp:485-494)
    {
        const uint32_t Ys = YK + 1u, Yb = 1u << (2 * (a.Kbg + 1));
        const float* vK = v + voff(K);
        const float* b = a.vbg + (((size_t)Yb - 4) / 3);
        for (uint32_t i = tid; i < W * Ys; i += nt) {
            const uint32_t j = i / Ys, y = i % Ys;
            a.s[i] = (y == YK) ? 1.0f : vK[(size_t)y * W + j] / b[y % Yb];
        }
    }
    if (tid == 0) {
        const double llh = stat3[3] != 0.0 ? (double)NAN : stat3[0], sum_r = stat3[1];
        const double nseq = a.n_seqs_override > 0.0 ? a.n_seqs_override : stat3[2];
        const uint32_t it = *a.iteration + 1u;
        *a.iteration = it;
        float q = *a.q;
        if (a.optimize_q)                                  // EM.cpp:515; the host applies EM.cpp:99's `iteration <= 5`
            q = (float)((nseq - sum_r + 1.0) / (nseq + 2.0));
        *a.q_out = q;
        if (a.stop != nullptr) {                           // EM.cpp:117-118
            const float llh_prev = a.llh_prev_from_status ? *a.llh_in : a.llh_prev;
            if ((float)v_diff < a.epsilon || ((float)llh - llh_prev < 0 && a.opt_iteration > 10u)) *a.stop = 1u;
        }
        if (a.llh_out != nullptr) *a.llh_out = (float)llh;
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
    }
}

// ---- model update spread over blocks (tables beyond the LDS form: K >= 3 at usual widths) ----------------------------
// One block walking 41 k cells order by order took 86 us of config 4's 6 ms iteration (1.4 %) and grows with 4^K.  Two
// launches instead, every cell computed on its own from what the launch before left in global memory:
//   k_update_counts  n_K from the integer accumulator; every lower-order cell summed straight from the accumulator's
//                    leaves in the reference's nesting (EM.cpp:247-254: four rows of the next order, ascending, from 0.0f)
//   k_update_model   per cell of every order the v chain from order 0 up (Motif.h:100-135, the expressions of
//                    model_update_lds), v_diff (per-block fp64 partials, summed in block order by the last block to
//                    draw a ticket: the same bits every run), the odds table; the accumulator is cleared here
// sum of the 4^D leaves under `row` in the reference's nesting: each level adds its four children in ascending order
template <int D, class Leaf>
__device__ __forceinline__ float count_tree(uint32_t row, uint32_t stride, const Leaf& leaf) {
    if constexpr (D == 0) return leaf(row);
    else {
        float s = 0.0f;
#pragma unroll
        for (uint32_t d = 0; d < 4; d++) s += count_tree<D - 1>(row + d * stride, stride * 4u, leaf);
        return s;
    }
}

// One band of orders: the cells of orders k_src-1 and k_src-2 summed from order k_src (FROM_ACC: the accumulator,
// k_src == K, whose own cells are converted here too; else the floats the band before wrote): at most 16 loads per
// thread, all issued before the first add.
template <bool FROM_ACC>
__global__ void __launch_bounds__(256) k_update_counts(UpdateArgs a, uint32_t k_src) {
    if (a.stop != nullptr && *a.stop != 0u) return;      // optimize(): the stop rule fired in an earlier pass
    const uint32_t W = a.W;
    auto voff = [W](uint32_t k) { return (size_t)W * (((size_t(1) << (2 * (k + 1))) - 4) / 3); };
    const uint32_t k_lo = k_src > 2u ? k_src - 2u : 0u;
    const size_t first = voff(k_lo), last = voff(FROM_ACC ? k_src + 1u : k_src);
    const float* const src = a.n + voff(k_src);
    for (size_t c = first + (size_t)blockIdx.x * blockDim.x + threadIdx.x; c < last; c += (size_t)gridDim.x * blockDim.x) {
        uint32_t k = k_lo;
        while (c >= voff(k + 1)) k++;
        const uint32_t i = (uint32_t)(c - voff(k)), y = i / W, j = i % W;
        auto leaf = [&](uint32_t row) {
            if constexpr (FROM_ACC) return (float)((double)a.acc[(size_t)row * W + j] * a.count_unit);
            else return src[(size_t)row * W + j];
        };
        const uint32_t Yk = 1u << (2 * (k + 1));
        const uint32_t depth = k_src - k;
        a.n[c] = depth == 0u ? leaf(y) : (depth == 1u ? count_tree<1>(y, Yk, leaf) : count_tree<2>(y, Yk, leaf));
    }
}

__global__ void __launch_bounds__(256) k_update_model(UpdateArgs a) {
    if (a.stop != nullptr && *a.stop != 0u) return;      // optimize(): the stop rule fired in an earlier pass
    __shared__ double shd[4];
    __shared__ uint32_t my_ticket;
    const uint32_t K = a.K, W = a.W;
    const uint32_t YK = 1u << (2 * (K + 1)), Ys = YK + 1u, Yb = 1u << (2 * (a.Kbg + 1));
    auto voff = [W](uint32_t k) { return (size_t)W * (((size_t(1) << (2 * (k + 1))) - 4) / 3); };
    const float* const n = a.n;
    const float* const b = a.vbg + (((size_t)Yb - 4) / 3);
    constexpr uint32_t kMaxK = 10;                           // bamm_em_create's limit
    double diff = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)YK * W; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t y = (uint32_t)(i / W), j = (uint32_t)(i % W);
        // every operand of the chain (Motif.h:100-135, as model_update_lds walks it) is addressed by (y, j) alone:
        // all loads first, then the dependent divisions
        float n0[4], num[kMaxK + 1], den[kMaxK + 1], Ak[kMaxK + 1];
#pragma unroll
        for (uint32_t yy = 0; yy < 4; yy++) n0[yy] = n[yy * W + j];
        Ak[0] = a.A[j];
        const float bg0 = a.vbg[y & 3u];
#pragma unroll
        for (uint32_t kk = 1; kk <= kMaxK; kk++) {
            if (kk > K || j < kk) continue;
            const uint32_t ykk = y & ((1u << (2 * (kk + 1))) - 1u);
            Ak[kk] = a.A[kk * W + j];
            num[kk] = n[voff(kk) + (size_t)ykk * W + j];
            den[kk] = n[voff(kk - 1) + (size_t)(ykk >> 2) * W + j - 1u];
        }
        const float old = a.v[voff(K) + i];                  // before the stores: v is updated in place
        const float bgK = b[y % Yb];
        float val = (n0[y & 3u] + Ak[0] * bg0) / (((n0[0] + n0[1]) + n0[2]) + n0[3] + Ak[0]);
        if (K > 0u && y < 4u) a.v[i] = val;                  // the lower orders' cells ride on the rows that spell them
#pragma unroll
        for (uint32_t kk = 1; kk <= kMaxK; kk++) {
            if (kk > K) continue;
            if (j >= kk) val = (num[kk] + Ak[kk] * val) / (den[kk] + Ak[kk]);
            if (kk < K && y < (1u << (2 * (kk + 1)))) a.v[voff(kk) + i] = val;
        }
        diff += (double)fabsf(val - old);
        a.v[voff(K) + i] = val;
        a.s[(size_t)j * Ys + y] = val / bgK;                 // Motif.cpp:485-494
        a.acc[i] = 0ll;                                      // consumed by k_update_counts
        if (a.acc_zero != nullptr) a.acc_zero[i] = 0ll;
        if (i < W) a.s[(size_t)i * Ys + YK] = 1.0f;         // the neutral row
    }
    // v_diff: wave sums, block sum, one partial per block
    {
        double d = diff;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) d += __shfl_xor(d, o, 64);
        if ((threadIdx.x & 63u) == 0u) shd[threadIdx.x >> 6] = d;
        __syncthreads();
        if (threadIdx.x == 0) {
            a.partial[blockIdx.x] = (shd[0] + shd[1]) + (shd[2] + shd[3]);
            __threadfence();
            my_ticket = atomicAdd(a.ticket, 1u);
        }
        __syncthreads();
    }
    if (my_ticket + 1u != gridDim.x) return;
    // the last block to finish: every partial is in memory (fence + atomic above); summed in an order that depends on
    // the grid alone
    __threadfence();
    {
        double d = 0.0;
        for (uint32_t bl = threadIdx.x; bl < gridDim.x; bl += blockDim.x)
            d += __hip_atomic_load(a.partial + bl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) d += __shfl_xor(d, o, 64);
        __syncthreads();                                     // shd was read by thread 0 above
        if ((threadIdx.x & 63u) == 0u) shd[threadIdx.x >> 6] = d;
        __syncthreads();
    }
    if (threadIdx.x != 0) return;
    *a.ticket = 0u;
    const double v_diff = (shd[0] + shd[1]) + (shd[2] + shd[3]);
    long long* const st = a.acc + (size_t)YK * W;
    const long long x0 = st[0], x1 = st[1], x2 = st[2];
    st[0] = 0ll; st[1] = 0ll; st[2] = 0ll;
    if (a.acc_zero != nullptr) { a.acc_zero[(size_t)YK * W] = 0ll; a.acc_zero[(size_t)YK * W + 1] = 0ll; a.acc_zero[(size_t)YK * W + 2] = 0ll; }
    const double llh = stat_bad(x2) ? (double)NAN : (double)x0 / kLlhScale, sum_r = (double)x1 / kSumrScale;
    const double nseq = a.n_seqs_override > 0.0 ? a.n_seqs_override : stat_nseq(x2);
    const uint32_t it = *a.iteration + 1u;
    *a.iteration = it;
    float q = *a.q;
    if (a.optimize_q)                                      // EM.cpp:515; the host applies EM.cpp:99's `iteration <= 5`
        q = (float)((nseq - sum_r + 1.0) / (nseq + 2.0));
    *a.q_out = q;
    if (a.stop != nullptr) {                               // EM.cpp:117-118
        const float llh_prev = a.llh_prev_from_status ? *a.llh_in : a.llh_prev;
        if ((float)v_diff < a.epsilon || ((float)llh - llh_prev < 0 && a.opt_iteration > 10u)) *a.stop = 1u;
    }
    if (a.llh_out != nullptr) *a.llh_out = (float)llh;
    a.status[0] = (float)llh; a.status[1] = (float)v_diff; a.status[2] = q; a.status[3] = (float)it;
    a.status[4] = (float)sum_r; a.status[5] = (float)nseq;
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

// EStep() alone: publish the statistics and clear them (the counts part was not touched)
__global__ void k_stat_only(long long* acc, uint32_t cells, float* status) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        status[0] = (float)((double)acc[cells + 0] / kLlhScale);
        status[4] = (float)((double)acc[cells + 1] / kSumrScale);
        status[5] = (float)stat_nseq(acc[cells + 2]);
        if (stat_bad(acc[cells + 2])) status[0] = NAN;
        acc[cells + 0] = 0ll; acc[cells + 1] = 0ll; acc[cells + 2] = 0ll;
    }
}

}  // namespace

int launch_reduce_partials(const unsigned long long* partial_n, const double* partial_stat, uint32_t blocks, uint32_t W,
                           uint32_t Y, long long* acc, hipStream_t st) {
    const uint32_t C = partial_n ? W * Y : 0u;
    const uint32_t grid = (C + 63u) / 64u + 1u;             // + the statistics block
    hipLaunchKernelGGL(k_reduce_partials, dim3(grid), dim3(1024), 0, st, partial_n, partial_stat, blocks, W, Y, acc);
    BAMM_HIP(hipGetLastError());
    return BAMM_OK;
}

// this translation unit's code object, loaded ahead of the first launch (bamm_ctx_create)
int prime_model_kernels() { return prime_kernel(reinterpret_cast<const void*>(&k_make_s)); }

int launch_make_s(const float* v, const float* vbg, uint32_t K, uint32_t W, uint32_t Kbg, float* s, hipStream_t st) {
    const uint32_t total = W * ((1u << (2 * (K + 1))) + 1u);
    hipLaunchKernelGGL(k_make_s, dim3((total + 255u) / 256u), dim3(256), 0, st, v, vbg, K, W, Kbg, s);
    BAMM_HIP(hipGetLastError());
    return BAMM_OK;
}

int launch_update(const UpdateArgs& a, hipStream_t st) {
    // n and v of all orders staged in LDS when at most two cells per thread of the top order are in flight
    // (the old v[K] is kept in two registers) and the tables fit the default 64 KiB
    if (update_fits_lds(a.K, a.W)) {
        hipLaunchKernelGGL(k_update<true>, dim3(1), dim3(1024), update_lds_bytes(a.K, a.W), st, a);
    } else if (a.partial != nullptr && a.ticket != nullptr) {
        const size_t cells = ipow4(a.K + 1) * a.W;
        const uint32_t b2 = (uint32_t)std::min<size_t>(kUpdateMaxBlocks, (cells + 255) / 256);
        for (uint32_t k_src = a.K;;) {                       // bands of two orders below their source
            const uint32_t k_lo = k_src > 2u ? k_src - 2u : 0u;
            const bool from_acc = k_src == a.K;
            const size_t band = v_offset(from_acc ? k_src + 1u : k_src, a.W) - v_offset(k_lo, a.W);
            const uint32_t b1 = (uint32_t)std::min<size_t>(kUpdateMaxBlocks, (band + 255) / 256);
            if (from_acc) hipLaunchKernelGGL(k_update_counts<true>, dim3(b1), dim3(256), 0, st, a, k_src);
            else hipLaunchKernelGGL(k_update_counts<false>, dim3(b1), dim3(256), 0, st, a, k_src);
            if (k_lo == 0u) break;
            k_src = k_lo;
        }
        hipLaunchKernelGGL(k_update_model, dim3(b2), dim3(256), 0, st, a);
    } else {
        hipLaunchKernelGGL(k_update<false>, dim3(1), dim3(1024), 0, st, a);
    }
    BAMM_HIP(hipGetLastError());
    return BAMM_OK;
}

int launch_stat_only(long long* acc, uint32_t cells, float* status, hipStream_t st) {
    hipLaunchKernelGGL(k_stat_only, dim3(1), dim3(64), 0, st, acc, cells, status);
    BAMM_HIP(hipGetLastError());
    return BAMM_OK;
}

}  // namespace bamm
