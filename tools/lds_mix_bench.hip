// LDS ceiling for the fused EM kernel's own instruction mix (figure iii).
//
// k_em_grp on the bench workload (K=2, W=20: 3 columns per table row, T=7 groups, 7 positions per lane)
// issues per sequence 14 `ds_read_b128` (7 positions x 2 quads of a random 48-byte table row) and 49
// predicated `ds_add_u64` (7 positions x 7 groups, ~15 of 64 lanes active, random rows of a [row][group]
// table).  This loop issues exactly that mix -- same table geometry, same 16 waves per CU, random rows,
// no decode, no chain arithmetic, no normalisation -- so its rate is what the LDS pipe alone allows for
// THIS access pattern (bank conflicts of random rows included).  Reads-only and adds-only rates are
// printed beside it.  One JSON line at the end for tools/summarize_pmc.py.
//   hipcc --offload-arch=gfx950 -O3 tools/lds_mix_bench.hip -o tools/lds_mix_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int M = 7, T = 7, ROWS = 1100, ROWSTRIDE_B = 48;      // bytes per odds-table row (3 quads)

template <int OFF>
__device__ __forceinline__ f32x4 rd128(uint32_t a) {
    f32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(OFF));
    return v;
}
template <int OFF>
__device__ __forceinline__ void add64(uint32_t a, unsigned long long v, unsigned long long mask) {
    unsigned long long saved;
    asm volatile("s_and_saveexec_b64 %0, %3\n\tds_add_u64 %1, %2 offset:%4\n\ts_mov_b64 exec, %0"
                 : "=&s"(saved) : "v"(a), "v"(v), "s"(mask), "n"(OFF) : "memory", "scc");
}

template <bool READS, bool ADDS>
__global__ void __launch_bounds__(1024) k_mix(int iters, int active_per_64, float* sink) {
    extern __shared__ __align__(16) unsigned char lds[];
    const uint32_t sg_bytes = ROWS * ROWSTRIDE_B, ng_off = (sg_bytes + 15u) & ~15u;
    for (uint32_t i = threadIdx.x; i < (ng_off + ROWS * T * 8u) / 4u; i += blockDim.x) reinterpret_cast<float*>(lds)[i] = 1.0f;
    __syncthreads();
    const uint32_t sg_base = (uint32_t)(size_t)(const __attribute__((address_space(3))) void*)lds;
    const uint32_t ng_base = sg_base + ng_off;
    uint32_t x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    float acc = 0.0f;
    for (int it = 0; it < iters; it++) {
        uint32_t row[M];
#pragma unroll
        for (int m = 0; m < M; m++) { x = x * 1664525u + 1013904223u; row[m] = (x >> 8) % ROWS; }
        if (READS) {
            f32x4 v[2 * M];
#pragma unroll
            for (int m = 0; m < M; m++) v[m] = rd128<0>(sg_base + row[m] * ROWSTRIDE_B);
#pragma unroll
            for (int m = 0; m < M; m++) v[M + m] = rd128<16>(sg_base + row[m] * ROWSTRIDE_B);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]));
#pragma unroll
            for (int m = 1; m < 2 * M; m++) asm volatile("" : "+v"(v[m]) : "v"(v[0]));
#pragma unroll
            for (int m = 0; m < 2 * M; m++) acc += v[m].x;
        }
        if (ADDS) {
            unsigned long long mask[M];
#pragma unroll
            for (int m = 0; m < M; m++) {
                x = x * 1664525u + 1013904223u;
                mask[m] = __ballot((int)((x >> 10) & 63u) < active_per_64);
            }
#pragma unroll
            for (int m = 0; m < M; m++) {
                const uint32_t a = ng_base + row[m] * (T * 8u);
                add64<0>(a, 3ull, mask[m]);  add64<8>(a, 3ull, mask[(m + 1) % M]);  add64<16>(a, 3ull, mask[(m + 2) % M]);
                add64<24>(a, 3ull, mask[(m + 3) % M]); add64<32>(a, 3ull, mask[(m + 4) % M]); add64<40>(a, 3ull, mask[(m + 5) % M]);
                add64<48>(a, 3ull, mask[(m + 6) % M]);
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (acc == 12345.678f) sink[0] = acc;
}


// The LDS traffic of a COMPACTED M-step (not built into the kernel unless this wins): the
// non-zero windows of a sequence (nnz of them, ~105 on the bench data) leave the lanes as (end position, r) pairs
// in a per-wave list, every lane of a round takes one listed window, re-derives its T rows from a per-wave copy
// of the 2-bit stream (26 words: broadcast reads, no bank conflicts) and issues T full adds.  Same 14 gathers.
__global__ void __launch_bounds__(1024) k_mix_list(int iters, int active_per_64, int nnz, float* sink) {
    extern __shared__ __align__(16) unsigned char lds[];
    const uint32_t sg_bytes = ROWS * ROWSTRIDE_B, ng_off = (sg_bytes + 15u) & ~15u;
    const uint32_t wave_off = ng_off + ROWS * T * 8u;                  // per wave: list 256 x 8 B, stream 32 words, region 64 floats
    for (uint32_t i = threadIdx.x; i < (wave_off + 16u * (2048u + 128u + 256u)) / 4u; i += blockDim.x) reinterpret_cast<float*>(lds)[i] = 1.0f;
    __syncthreads();
    const uint32_t sg_base = (uint32_t)(size_t)(const __attribute__((address_space(3))) void*)lds;
    const uint32_t ng_base = sg_base + ng_off;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t list_base = sg_base + wave_off + wave * (2048u + 128u + 256u), str_base = list_base + 2048u, reg_base = str_base + 128u;
    uint32_t x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    float acc = 0.0f;
    for (int it = 0; it < iters; it++) {
        uint32_t row[M];
#pragma unroll
        for (int m = 0; m < M; m++) { x = x * 1664525u + 1013904223u; row[m] = (x >> 8) % ROWS; }
        {
            f32x4 v[2 * M];
#pragma unroll
            for (int m = 0; m < M; m++) v[m] = rd128<0>(sg_base + row[m] * ROWSTRIDE_B);
#pragma unroll
            for (int m = 0; m < M; m++) v[M + m] = rd128<16>(sg_base + row[m] * ROWSTRIDE_B);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]));
#pragma unroll
            for (int m = 1; m < 2 * M; m++) asm volatile("" : "+v"(v[m]) : "v"(v[0]));
#pragma unroll
            for (int m = 0; m < 2 * M; m++) acc += v[m].x;
        }
        // stream copy: lanes 0..25 one word each
        if (lane < 26u) asm volatile("ds_write_b32 %0, %1" :: "v"(str_base + lane * 4u), "v"(x) : "memory");
        // list: 7 predicated 8-byte writes, compacted (rank within the slot + running count)
        uint32_t cnt = 0;
#pragma unroll
        for (int m = 0; m < M; m++) {
            x = x * 1664525u + 1013904223u;
            const bool on = (int)((x >> 10) & 63u) < active_per_64;
            const unsigned long long mk = __ballot(on);
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, 0u));
            if (on && cnt + rank < 256u) asm volatile("ds_write_b64 %0, %1" :: "v"(list_base + (cnt + rank) * 8u), "v"((unsigned long long)x) : "memory");
            cnt += (uint32_t)__builtin_popcountll(mk);
            // region copy of r for the fix lanes: 8 lanes
            if ((lane & 7u) == 3u && lane < 64u) asm volatile("ds_write_b32 %0, %1" :: "v"(reg_base + (lane >> 3) * 28u + m * 4u), "v"(x) : "memory");
        }
        // rounds over the listed windows (nnz fixed by the caller so that runs are comparable)
        for (int base = 0; base < nnz; base += 64) {
            const bool on = (int)lane + base < nnz;
            if (on) {
                unsigned long long e;
                uint32_t w0, w1, w2;
                const uint32_t k = (x >> 7) % 24u;
                asm volatile("ds_read_b64 %0, %1" : "=v"(e) : "v"(list_base + ((lane + base) & 255u) * 8u));
                asm volatile("ds_read_b32 %0, %1" : "=v"(w0) : "v"(str_base + k * 4u));
                asm volatile("ds_read_b32 %0, %1 offset:4" : "=v"(w1) : "v"(str_base + k * 4u));
                asm volatile("ds_read_b32 %0, %1 offset:8" : "=v"(w2) : "v"(str_base + k * 4u));
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(e), "+v"(w0), "+v"(w1), "+v"(w2));
                x ^= (uint32_t)e ^ w0 ^ w1 ^ w2;
                uint32_t h = x;
#pragma unroll
                for (int u = 0; u < T; u++) {
                    h = h * 1664525u + 1013904223u;
                    const uint32_t a = ng_base + ((h >> 8) % ROWS) * (T * 8u) + u * 8u;
                    asm volatile("ds_add_u64 %0, %1" :: "v"(a), "v"(3ull) : "memory");
                }
            }
        }
        // fix lanes: two reads of the region, three adds into single-column bins
        if (lane < 42u) {
            uint32_t r0, r1;
            asm volatile("ds_read_b32 %0, %1" : "=v"(r0) : "v"(reg_base + (lane % 56u) * 4u));
            asm volatile("ds_read_b32 %0, %1 offset:4" : "=v"(r1) : "v"(reg_base + (lane % 56u) * 4u));
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r0), "+v"(r1));
            const uint32_t a = ng_base + ((r0 ^ r1 ^ lane * 977u) % (ROWS * T)) * 8u;
            asm volatile("ds_add_u64 %0, %1" :: "v"(a), "v"(1ull) : "memory");
            asm volatile("ds_add_u64 %0, %1 offset:8" :: "v"(a), "v"(1ull) : "memory");
            asm volatile("ds_add_u64 %0, %1 offset:16" :: "v"(a), "v"(1ull) : "memory");
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (acc == 12345.678f) sink[0] = acc;
}

int run_list(int iters, int active, int nnz, float* sink) {
    const size_t lds = ((ROWS * ROWSTRIDE_B + 15) & ~15) + ROWS * T * 8 + 16 * (2048 + 128 + 256);
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_mix_list), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_mix_list, dim3(256), dim3(1024), lds, 0, iters / 10, active, nnz, sink);
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k_mix_list, dim3(256), dim3(1024), lds, 0, iters, active, nnz, sink);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("compacted M-step, nnz %3d       %8.3f ms  %6.2f ns per sequence-equivalent per CU\n", nnz, ms, ms * 1e6 / (16.0 * iters));
    return 0;
}

// A mixed-row formulation (not built): W = 20 at K = 2 as groups of 4+4+3+3+3+3 columns -- two groups on 6-mer rows
// (4096 rows: odds [4096][2] floats, counts [4096][2] u64), four on 5-mer rows (odds [1100][4] floats, counts
// [1100][5] u64, odd stride) -- 6 adds and one b64 + one b128 gather per position instead of 7 and two b128.
__global__ void __launch_bounds__(1024) k_mix_66(int iters, int active_per_64, float* sink) {
    extern __shared__ __align__(16) unsigned char lds[];
    constexpr uint32_t R6 = 4096u + 128u, R5 = 1100u;
    const uint32_t o6 = 0, o5 = o6 + R6 * 8u, c6 = o5 + R5 * 16u, c5 = c6 + R6 * 16u, total = c5 + R5 * 40u;
    for (uint32_t i = threadIdx.x; i < total / 4u; i += blockDim.x) reinterpret_cast<float*>(lds)[i] = 1.0f;
    __syncthreads();
    const uint32_t base = (uint32_t)(size_t)(const __attribute__((address_space(3))) void*)lds;
    uint32_t x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    float acc = 0.0f;
    for (int it = 0; it < iters; it++) {
        uint32_t r6[M], r5[M];
#pragma unroll
        for (int m = 0; m < M; m++) { x = x * 1664525u + 1013904223u; r6[m] = (x >> 8) & 4095u; r5[m] = (x >> 20) % R5; }
        {
            f32x4 v[M];
            float2 w[M];
#pragma unroll
            for (int m = 0; m < M; m++) asm volatile("ds_read_b64 %0, %1" : "=v"(w[m]) : "v"(base + o6 + r6[m] * 8u));
#pragma unroll
            for (int m = 0; m < M; m++) v[m] = rd128<0>(base + o5 + r5[m] * 16u);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]));
#pragma unroll
            for (int m = 0; m < M; m++) asm volatile("" : "+v"(v[m]), "+v"(w[m]) : "v"(v[0]));
#pragma unroll
            for (int m = 0; m < M; m++) acc += v[m].x + w[m].x;
        }
        unsigned long long mask[M];
#pragma unroll
        for (int m = 0; m < M; m++) { x = x * 1664525u + 1013904223u; mask[m] = __ballot((int)((x >> 10) & 63u) < active_per_64); }
#pragma unroll
        for (int m = 0; m < M; m++) {
            const uint32_t a6 = base + c6 + r6[m] * 16u, a5 = base + c5 + r5[m] * 40u;
            add64<0>(a6, 3ull, mask[m]); add64<8>(a6, 3ull, mask[(m + 1) % M]);
            add64<0>(a5, 3ull, mask[(m + 2) % M]); add64<8>(a5, 3ull, mask[(m + 3) % M]);
            add64<16>(a5, 3ull, mask[(m + 4) % M]); add64<24>(a5, 3ull, mask[(m + 5) % M]);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (acc == 12345.678f) sink[0] = acc;
}

int run_66(int iters, int active, float* sink, double* rate_out) {
    const size_t lds = (4096 + 128) * 8 + 1100 * 16 + (4096 + 128) * 16 + 1100 * 40;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_mix_66), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_mix_66, dim3(256), dim3(1024), lds, 0, iters / 10, active, sink);
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k_mix_66, dim3(256), dim3(1024), lds, 0, iters, active, sink);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("mixed rows (4+4 on 6-mers, 3+3+3+3 on 5-mers): 7 b64 + 7 b128 gathers + 42 adds  %8.3f ms  %6.2f ns per sequence-equivalent per CU\n",
           ms, ms * 1e6 / (16.0 * iters));
    if (rate_out) *rate_out = 256.0 * 16.0 * (double)iters * 56.0 / (ms * 1e-3);
    return 0;
}

template <bool R, bool A>
int run(const char* name, int per_iter, int iters, int active, float* sink, double* rate_out) {
    const size_t lds = ((ROWS * ROWSTRIDE_B + 15) & ~15) + ROWS * T * 8;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_mix<R, A>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_mix<R, A>), dim3(256), dim3(1024), lds, 0, iters / 10, active, sink);
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((k_mix<R, A>), dim3(256), dim3(1024), lds, 0, iters, active, sink);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double wave_instr = 256.0 * 16.0 * (double)iters * per_iter;
    const double rate = wave_instr / (ms * 1e-3);
    printf("%-28s %8.3f ms  %6.2f ns per sequence-equivalent per CU  %.3e LDS wave-instr/s  (%.2f cycles per wave-instr per CU @2.4 GHz)\n",
           name, ms, ms * 1e6 / (16.0 * iters), rate, 2.4e9 * 256.0 / rate);
    if (rate_out) *rate_out = rate;
    return 0;
}

int main(int argc, char** argv) {
    const int iters = 4000, active = argc > 1 ? atoi(argv[1]) : 15;
    float* sink;
    CK(hipMalloc(&sink, 4));
    double mix = 0, r = 0, a = 0;
    if (run<true, false>("14 ds_read_b128", 14, iters, active, sink, &r)) return 1;
    if (run<false, true>("49 ds_add_u64", 49, iters, active, sink, &a)) return 1;
    if (run<true, true>("14 reads + 49 adds", 63, iters, active, sink, &mix)) return 1;
    if (argc > 2) {                                      // the compacted formulation's LDS traffic, for comparison
        const int nnzs[] = {64, 105, 128, 160, 192, 256};
        for (int nnz : nnzs) if (run_list(iters, active, nnz, sink)) return 1;
    }
    double mixed = 0;
    if (run_66(iters, active, sink, &mixed)) return 1;
    printf("{\"wave_instr_per_s\": %.6e, \"reads_only_wave_instr_per_s\": %.6e, \"adds_only_wave_instr_per_s\": %.6e, "
           "\"active_lanes_per_add\": %d, \"what\": \"LDS-only loop of k_em_grp's mix on the bench workload: 14 ds_read_b128 + 49 "
           "predicated ds_add_u64 per sequence, random rows of the same tables, 16 waves per CU (tools/lds_mix_bench.hip)\", "
           "\"mixed_rows_wave_instr_per_s\": %.6e, \"mixed_rows_what\": \"LDS-only loop of k_em_mix's mix on the bench workload: 7 ds_read_b64 "
           "+ 7 ds_read_b128 + 42 predicated ds_add_u64 per sequence, random rows of the same tables, 16 waves per CU "
           "(tools/lds_mix_bench.hip)\"}\n",
           mix, r, a, active, mixed);
    return 0;
}
