// gfx950 kernel of Motif::initFromPWM's pass over the sequences
// (/root/reference/src/init/Motif.cpp:228-311): the 0th-order posterior of every window, ONE sampled
// motif start per sequence and the integer k-mer counts of the sampled sites.
//
// The reference draws the start with std::discrete_distribution from a default-seeded std::mt19937,
// sequence after sequence (Motif.cpp:237,296-299).  Only the uniform variates are serial: the host
// draws them in sequence order (std::generate_canonical<double,53>, which is what the distribution
// calls) and everything else happens here, restating libstdc++'s arithmetic so that the sampled
// index is the one the reference gets:
//
//   r[i]  = (((1.0f * s[0][b(i-1)]) * s[1][b(i)]) ... ) * (q / LW1)           float, i = 1..LW1   (:277-287)
//   nf    = ((r[1] + r[2]) + ...) + (1 - q)                                     float, sequential   (:285,290)
//   r[i] /= nf                                                                  float               (:292-294)
//   sum   = (double)r[0] + (double)r[1] + ...          std::accumulate(..., 0.0)
//   p[i]  = (double)r[i] / sum;   cp[i] = cp[i-1] + p[i];   cp[LW1] = 1.0       discrete_distribution::_M_initialize
//   z     = first i with cp[i] >= u                                             std::lower_bound
//
// One wavefront per sequence.  The window products and the two element-wise divisions run across the
// lanes; the three order-dependent sums are walked by lane 0 over per-wave LDS arrays (four values per
// LDS round trip) -- ~10k cycles per sequence and wave, which 4096 resident waves turn into a
// millisecond per million sequences.  Nothing here is on the per-iteration path.
// Sequences beyond ~10 000 positions (16 bytes of per-wave arrays per position exceed the LDS): the same kernel
// with the arrays in a global scratch region per wave (WG) -- the reference has no limit (Motif.cpp:228-311).

#include "device_utils.h"

#include <algorithm>

namespace bamm {
namespace {

__device__ __forceinline__ void seed_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <bool WG>
__global__ void __launch_bounds__(1024) k_seed_pwm(SeedKernelArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    const uint32_t W = a.W, K = a.K, Y = a.Y;
    float* sc = reinterpret_cast<float*>(smem);                             // [4][W] odds of the floored PWM
    int* cnt = reinterpret_cast<int*>(smem + a.table_bytes);                // block-private counts (when they fit)
    const bool cnt_lds = a.count_bytes != 0u;
    for (uint32_t i = threadIdx.x; i < 4u * W; i += blockDim.x) sc[i] = a.score[i];
    if (cnt_lds)
        for (uint32_t i = threadIdx.x; i < a.vsize; i += blockDim.x) cnt[i] = 0;
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t waves_per_block = blockDim.x >> 6;
    const uint32_t total_waves = gridDim.x * waves_per_block;
    unsigned char* wbase;
    if constexpr (WG) wbase = a.wave_scratch + ((size_t)blockIdx.x * waves_per_block + wave) * a.wave_bytes;
    else wbase = smem + a.table_bytes + a.count_bytes + (size_t)wave * a.wave_bytes;
    const uint32_t n4 = (a.max_len * 4u + 4u + 15u) & ~15u, n8 = (a.max_len * 8u + 8u + 15u) & ~15u;
    uint32_t* ybuf = reinterpret_cast<uint32_t*>(wbase);                    // [max_len]   kmer_ mod 4^(K+1)
    float* rf = reinterpret_cast<float*>(wbase + n4);                       // [max_len+1] r
    double* pd = reinterpret_cast<double*>(wbase + 2 * (size_t)n4);         // [max_len+1] p
    (void)n8;
    const float pos0 = 1.0f - a.q;

    for (uint32_t t = blockIdx.x * waves_per_block + wave; t < a.sv.count; t += total_waves) {
        const uint32_t seq = pick_sequence(a.sv, t);
        const uint32_t L = a.sv.len[seq];
        if (L < W) { if (lane == 0 && a.z_out) a.z_out[seq] = 0u; continue; }          // Motif.cpp:240-248
        const uint32_t LW1 = L - W + 1u;
        // kmer_[p] mod 4^(K+1) for every position (Sequence.cpp:35-41), N exceptions applied
        {
            const uint32_t* wp = a.sv.words + a.sv.word_off[seq];
            for (uint32_t p = (uint32_t)lane; p < L; p += 64u) {
                const uint32_t wi = p >> 4;
                const uint32_t lo = wp[wi], hi = wi ? wp[wi - 1u] : 0u;
                ybuf[p] = __builtin_amdgcn_alignbit(hi, lo, 30u - 2u * (p & 15u)) & (Y - 1u);
            }
            seed_lds_sync();
            const uint64_t e0 = a.sv.exc_off[seq], e1 = a.sv.exc_off[seq + 1];
            for (uint64_t e = e0 + (uint64_t)lane; e < e1; e += 64u) {
                const uint2 x = a.sv.exc[e];
                if (x.x < L) ybuf[x.x] = x.y;
            }
            seed_lds_sync();
        }
        // window products, Motif.cpp:277-287 (same multiplication order; digit 0 of the k-mer is the base)
        const float pos1 = a.q / (float)LW1;
        for (uint32_t i = (uint32_t)lane; i < LW1; i += 64u) {
            float r = 1.0f;
            for (uint32_t j = 0; j < W; j++) r *= sc[(ybuf[i + j] & 3u) * W + j];
            rf[i + 1u] = r * pos1;
        }
        if (lane == 0) rf[0] = pos0;
        seed_lds_sync();
        float nf = 0.0f;
        if (lane == 0) {                                     // sequential fp32 sum, then + (1-q)  (:285,290)
            uint32_t i = 1;
            for (; i + 4u <= LW1 + 1u; i += 4u) { nf += rf[i]; nf += rf[i + 1]; nf += rf[i + 2]; nf += rf[i + 3]; }
            for (; i <= LW1; i++) nf += rf[i];
            nf += pos0;
        }
        nf = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, nf)));
        for (uint32_t i = (uint32_t)lane; i <= LW1; i += 64u) rf[i] = rf[i] / nf;       // :292-294
        seed_lds_sync();
        double sum = 0.0;
        if (lane == 0) {                                     // std::accumulate(begin, end, 0.0)
            uint32_t i = 0;
            for (; i + 4u <= LW1 + 1u; i += 4u) {
                sum += (double)rf[i]; sum += (double)rf[i + 1]; sum += (double)rf[i + 2]; sum += (double)rf[i + 3];
            }
            for (; i <= LW1; i++) sum += (double)rf[i];
        }
        {
            const unsigned long long sb = __builtin_bit_cast(unsigned long long, sum);
            const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)sb), hi = __builtin_amdgcn_readfirstlane((uint32_t)(sb >> 32));
            sum = __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
        }
        for (uint32_t i = (uint32_t)lane; i <= LW1; i += 64u) pd[i] = (double)rf[i] / sum;   // __normalize
        seed_lds_sync();
        uint32_t z = LW1;                                    // cp[LW1] is forced to 1.0 and u < 1
        if (lane == 0) {
            const double u = a.u[seq];
            double cp = 0.0;
            for (uint32_t i = 0; i < LW1; i++) {            // std::partial_sum + std::lower_bound
                cp += pd[i];
                if (cp >= u) { z = i; break; }
            }
        }
        z = __builtin_amdgcn_readfirstlane(z);
        if (lane == 0 && a.z_out) a.z_out[seq] = z;
        if (z > 0u) {                                        // Motif.cpp:302-309: counts of every order at the sampled site
            for (uint32_t j = (uint32_t)lane; j < W; j += 64u) {
                const uint32_t y = ybuf[z - 1u + j];
                uint32_t off = 0;
                for (uint32_t k = 0; k <= K; k++) {
                    const uint32_t Yk = 1u << (2u * (k + 1u));
                    const uint32_t cell = off + (y & (Yk - 1u)) * W + j;
                    if (cnt_lds) atomicAdd(&cnt[cell], 1); else atomicAdd(&a.counts[cell], 1);
                    off += Yk * W;
                }
            }
        }
        seed_lds_sync();
    }
    __syncthreads();
    if (cnt_lds)
        for (uint32_t i = threadIdx.x; i < a.vsize; i += blockDim.x)
            if (cnt[i] != 0) atomicAdd(&a.counts[i], cnt[i]);
}

}  // namespace

size_t seed_wave_bytes(uint32_t max_len) {
    const size_t n4 = ((size_t)max_len * 4 + 4 + 15) & ~size_t(15), n8 = ((size_t)max_len * 8 + 8 + 15) & ~size_t(15);
    return 2 * n4 + n8;
}

namespace {
struct SeedPlan { uint32_t table_bytes, count_bytes, wave_bytes, waves, blocks; size_t lds; bool global; };

SeedPlan seed_plan(const SeedKernelArgs& a, uint32_t num_cus) {
    const size_t kLds = 160 * 1024;
    SeedPlan p{};
    p.table_bytes = (uint32_t)(((size_t)4 * a.W * sizeof(float) + 15) & ~size_t(15));
    const size_t wave_bytes = seed_wave_bytes(a.max_len);
    p.wave_bytes = (uint32_t)wave_bytes;
    const size_t want_counts = ((size_t)a.vsize * sizeof(int) + 15) & ~size_t(15);
    p.count_bytes = want_counts <= 48 * 1024 ? (uint32_t)want_counts : 0u;
    if ((size_t)p.table_bytes + p.count_bytes + wave_bytes > kLds) p.count_bytes = 0u;   // long sequences: counts straight to HBM
    p.global = (size_t)p.table_bytes + p.count_bytes + wave_bytes > kLds;
    const uint32_t cus = num_cus ? num_cus : 256u;
    if (p.global) {
        // per-wave arrays in global memory: the block's counts are back in LDS, at most 4 GiB of arrays in all
        p.count_bytes = want_counts <= 48 * 1024 ? (uint32_t)want_counts : 0u;
        const size_t max_waves = std::max<size_t>(1, (size_t(4) << 30) / wave_bytes);
        p.waves = (uint32_t)std::min<size_t>(4, max_waves);
        p.blocks = (uint32_t)std::max<size_t>(1, std::min<size_t>({(size_t)cus, (a.sv.count + p.waves - 1u) / p.waves, max_waves / p.waves}));
        p.lds = (size_t)p.table_bytes + p.count_bytes;
    } else {
        const size_t fixed = (size_t)p.table_bytes + p.count_bytes;
        p.waves = (uint32_t)std::min<size_t>(16, (kLds - fixed) / wave_bytes);
        p.lds = fixed + (size_t)p.waves * wave_bytes;
        p.blocks = std::max(1u, std::min(cus, (a.sv.count + p.waves - 1u) / p.waves));
    }
    return p;
}
}  // namespace

size_t seed_global_scratch_bytes(const SeedKernelArgs& a, uint32_t num_cus) {
    const SeedPlan p = seed_plan(a, num_cus);
    return p.global ? (size_t)p.blocks * p.waves * seed_wave_bytes(a.max_len) : 0;
}

int launch_seed_pwm(SeedKernelArgs a, uint32_t num_cus, hipStream_t st) {
    if (seed_wave_bytes(a.max_len) > 0xffffffffull) {
        set_error("PWM seeding: a sequence of %u positions exceeds the per-wave arrays' 4 GiB", a.max_len);
        return BAMM_ERR_UNSUPPORTED;
    }
    const SeedPlan p = seed_plan(a, num_cus);
    a.table_bytes = p.table_bytes; a.count_bytes = p.count_bytes; a.wave_bytes = p.wave_bytes;
    if (p.global) {
        if (a.wave_scratch == nullptr) { set_error("PWM seeding: no scratch for the per-wave arrays"); return BAMM_ERR_ARG; }
        if (int rc = allow_lds(reinterpret_cast<const void*>(&k_seed_pwm<true>), p.lds)) return rc;
        hipLaunchKernelGGL(k_seed_pwm<true>, dim3(p.blocks), dim3(p.waves * 64u), p.lds, st, a);
    } else {
        if (int rc = allow_lds(reinterpret_cast<const void*>(&k_seed_pwm<false>), p.lds)) return rc;
        hipLaunchKernelGGL(k_seed_pwm<false>, dim3(p.blocks), dim3(p.waves * 64u), p.lds, st, a);
    }
    BAMM_HIP(hipGetLastError());
    return BAMM_OK;
}

}  // namespace bamm
