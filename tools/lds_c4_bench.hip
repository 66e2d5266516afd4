// LDS ceilings for the two kernels of the column-sliced path on config 4 (1M x 500 bp both strands: L = 1001, 16 positions
// per lane; W = 30, k = 4: Y = 1024), SURVEY.md 8d figure (iii) -- what tools/lds_mix_bench.hip is for the fused kernels.
//
// E pass (k_em_seq, kernels.hip, odds table [8 quads][1025 rows][4] floats = 131 KB in LDS, 12 waves per CU): per
//   sequence 8 x 16 = 128 `ds_read_b128` of random rows, 16 in flight at a time.
// M slice (k_m_list, 15 columns of the count table [15][1025] u64 = 123 KB + a u16 copy of the decoded sequence per wave):
//   per sequence 16 `ds_write_b16`, then per round of 64 listed windows 15 `ds_read_u16` + 15 full-lane `ds_add_u64` on
//   random rows (`rounds` is the second argument; default 2: in steady state the PMC counters show 16 + ~1.7 x 30 LDS
//   wave-instructions per sequence and slice, i.e. ~110 of 972 windows listed).
// Same table geometry and waves per CU as the kernels, random rows, no decode, no chain, no HBM: the rate the LDS pipe alone
// allows for THESE access patterns.  One JSON line at the end for tools/summarize_pmc.py (--lds-c4).
//   hipcc --offload-arch=gfx950 -O3 tools/lds_c4_bench.hip -o tools/lds_c4_bench && tools/lds_c4_bench [iters] [rounds]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int M = 16, QUADS = 8, NC = 15, THREADS = 768;
constexpr uint32_t YS = 1025u;

__global__ void __launch_bounds__(THREADS) k_e(int iters, float* sink) {
    extern __shared__ __align__(16) unsigned char lds[];
    for (uint32_t i = threadIdx.x; i < QUADS * YS * 4u; i += blockDim.x) reinterpret_cast<float*>(lds)[i] = 1.0f;
    __syncthreads();
    const uint32_t base = (uint32_t)(size_t)(const __attribute__((address_space(3))) void*)lds;
    uint32_t x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    float acc = 0.0f;
    for (int it = 0; it < iters; it++) {
        uint32_t a[M];
#pragma unroll
        for (int m = 0; m < M; m++) { x = x * 1664525u + 1013904223u; a[m] = base + ((x >> 8) & 1023u) * 16u; }
#pragma unroll
        for (int q = 0; q < QUADS; q++) {
            f32x4 v[M];
#pragma unroll
            for (int m = 0; m < M; m++) asm volatile("ds_read_b128 %0, %1" : "=v"(v[m]) : "v"(a[m] + (uint32_t)q * YS * 16u));
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]));
#pragma unroll
            for (int m = 1; m < M; m++) asm volatile("" : "+v"(v[m]) : "v"(v[0]));
#pragma unroll
            for (int m = 0; m < M; m++) acc += v[m].x;
        }
    }
    if (acc == 12345.678f) sink[0] = acc;
}

__global__ void __launch_bounds__(THREADS) k_m(int iters, int rounds, float* sink) {
    extern __shared__ __align__(16) unsigned char lds[];
    constexpr uint32_t table = NC * YS * 8u, ybytes = (64u * M + 8u) * 2u;
    for (uint32_t i = threadIdx.x; i < (table + (THREADS / 64) * ybytes) / 4u; i += blockDim.x) reinterpret_cast<uint32_t*>(lds)[i] = 0u;
    __syncthreads();
    const uint32_t base = (uint32_t)(size_t)(const __attribute__((address_space(3))) void*)lds;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t ybuf = base + table + wave * ybytes;
    uint32_t x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    uint32_t acc = 0;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int m = 0; m < M; m++) {
            x = x * 1664525u + 1013904223u;
            asm volatile("ds_write_b16 %0, %1" :: "v"(ybuf + (lane * M + m) * 2u), "v"((x >> 8) & 1023u) : "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        for (int r = 0; r < rounds; r++) {
            x = x * 1664525u + 1013904223u;
            const uint32_t q = (x >> 9) % (64u * M - 30u);              // the listed window's first column position
            for (int jb = 0; jb < NC; jb += 8) {
                uint32_t yy[8];
#pragma unroll
                for (int c = 0; c < 8; c++)
                    if (jb + c < NC) asm volatile("ds_read_u16 %0, %1" : "=v"(yy[c]) : "v"(ybuf + (q + jb + c) * 2u));
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(yy[0]));
#pragma unroll
                for (int c = 0; c < 8; c++)
                    if (jb + c < NC) {
                        asm volatile("" : "+v"(yy[c]) : "v"(yy[0]));
                        acc += yy[c];
                        asm volatile("ds_add_u64 %0, %1" :: "v"(base + ((uint32_t)(jb + c) * YS + (yy[c] & 1023u)) * 8u), "v"(3ull) : "memory");
                    }
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (acc == 0x12345678u) sink[0] = (float)acc;
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 2000, rounds = argc > 2 ? atoi(argv[2]) : 2;
    float* sink;
    CK(hipMalloc(&sink, 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t lds_e = QUADS * YS * 16, lds_m = NC * YS * 8 + (THREADS / 64) * (64 * M + 8) * 2;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_e), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_e));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_m), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_m));
    float ms;
    hipLaunchKernelGGL(k_e, dim3(256), dim3(THREADS), lds_e, 0, iters / 10, sink);
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k_e, dim3(256), dim3(THREADS), lds_e, 0, iters, sink);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double waves = 256.0 * (THREADS / 64);
    const double e_rate = waves * iters * (double)(QUADS * M) / (ms * 1e-3);
    printf("E pass: 128 ds_read_b128 per sequence          %8.3f ms  %7.1f ns per sequence per CU  %.3e wave-instr/s (%.2f cycles each per CU @2.4 GHz)\n",
           ms, ms * 1e6 / ((THREADS / 64) * (double)iters), e_rate, 2.4e9 * 256.0 / e_rate);
    hipLaunchKernelGGL(k_m, dim3(256), dim3(THREADS), lds_m, 0, iters / 10, rounds, sink);
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k_m, dim3(256), dim3(THREADS), lds_m, 0, iters, rounds, sink);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double per_seq = M + rounds * 2.0 * NC;
    const double m_rate = waves * iters * per_seq / (ms * 1e-3);
    printf("M slice: 16 writes + %d x (15 u16 reads + 15 adds) %8.3f ms  %7.1f ns per sequence per CU  %.3e wave-instr/s (%.2f cycles each per CU @2.4 GHz)\n",
           rounds, ms, ms * 1e6 / ((THREADS / 64) * (double)iters), m_rate, 2.4e9 * 256.0 / m_rate);
    printf("{\"e_pass_wave_instr_per_s\": %.6e, \"m_list_wave_instr_per_s\": %.6e, \"rounds\": %d, \"what\": \"LDS-only loops of config 4's kernels "
           "(tools/lds_c4_bench.hip): E pass 128 ds_read_b128 of random rows of the 131 KB odds table per sequence; M slice 16 ds_write_b16 + "
           "rounds x (15 ds_read_u16 + 15 full-lane ds_add_u64 on random rows of a 15-column count slice); 12 waves per CU\"}\n",
           e_rate, m_rate, rounds);
    return 0;
}
