// What a predicated LDS add costs by operand width and by whether it returns (gfx950).
//
// k_em_grp's M-step issues 49 predicated `ds_add_u64` per sequence, ~15 of 64 lanes active, and
// tools/lds_mix_bench.hip shows 6.4 LDS cycles per instruction however few lanes take part.  This loop
// asks whether a narrower or a returning add has a lower floor: the same 49 adds per iteration into a
// [row][group] table of random rows, 16 waves per CU, as
//   u64      ds_add_u64                       (what the kernel issues)
//   u32      ds_add_u32                       (low dword only)
//   u32x2    ds_add_u32 twice                 (low and high dword, no carry)
//   rtn32    ds_add_rtn_u32, 7 per wait       (low dword, old value back: the carry test)
//   rtn64    ds_add_rtn_u64, 7 per wait
//   hipcc --offload-arch=gfx950 -O3 tools/lds_add_width_bench.hip -o tools/lds_add_width_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int M = 7, T = 7, ROWS = 1100;

template <int OFF>
__device__ __forceinline__ void add64(uint32_t a, unsigned long long v, unsigned long long mask) {
    unsigned long long saved;
    asm volatile("s_and_saveexec_b64 %0, %3\n\tds_add_u64 %1, %2 offset:%4\n\ts_mov_b64 exec, %0"
                 : "=&s"(saved) : "v"(a), "v"(v), "s"(mask), "n"(OFF) : "memory", "scc");
}
template <int OFF>
__device__ __forceinline__ void add32(uint32_t a, uint32_t v, unsigned long long mask) {
    unsigned long long saved;
    asm volatile("s_and_saveexec_b64 %0, %3\n\tds_add_u32 %1, %2 offset:%4\n\ts_mov_b64 exec, %0"
                 : "=&s"(saved) : "v"(a), "v"(v), "s"(mask), "n"(OFF) : "memory", "scc");
}
template <int OFF>
__device__ __forceinline__ uint32_t add32r(uint32_t a, uint32_t v, unsigned long long mask) {
    unsigned long long saved;
    uint32_t old;
    asm volatile("s_and_saveexec_b64 %0, %4\n\tds_add_rtn_u32 %1, %2, %3 offset:%5\n\ts_mov_b64 exec, %0"
                 : "=&s"(saved), "=&v"(old) : "v"(a), "v"(v), "s"(mask), "n"(OFF) : "memory", "scc");
    return old;
}
template <int OFF>
__device__ __forceinline__ unsigned long long add64r(uint32_t a, unsigned long long v, unsigned long long mask) {
    unsigned long long saved, old;
    asm volatile("s_and_saveexec_b64 %0, %4\n\tds_add_rtn_u64 %1, %2, %3 offset:%5\n\ts_mov_b64 exec, %0"
                 : "=&s"(saved), "=&v"(old) : "v"(a), "v"(v), "s"(mask), "n"(OFF) : "memory", "scc");
    return old;
}

template <int MODE>
__global__ void __launch_bounds__(1024) k_add(int iters, int active_per_64, uint32_t* sink) {
    extern __shared__ __align__(16) unsigned char lds[];
    for (uint32_t i = threadIdx.x; i < ROWS * T * 2u; i += blockDim.x) reinterpret_cast<uint32_t*>(lds)[i] = 0u;
    __syncthreads();
    const uint32_t base = (uint32_t)(size_t)(const __attribute__((address_space(3))) void*)lds;
    uint32_t x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    uint32_t acc = 0u;
    for (int it = 0; it < iters; it++) {
        uint32_t row[M];
        unsigned long long mask[M];
#pragma unroll
        for (int m = 0; m < M; m++) {
            x = x * 1664525u + 1013904223u;
            row[m] = (x >> 8) % ROWS;
            mask[m] = __ballot((int)((x >> 3) & 63u) < active_per_64);
        }
#pragma unroll
        for (int m = 0; m < M; m++) {
            const uint32_t a = base + row[m] * (T * 8u);
#define ALL7(F) F(0, 0) F(8, 1) F(16, 2) F(24, 3) F(32, 4) F(40, 5) F(48, 6)
            if (MODE == 0) {
#define F(OFF, S) add64<OFF>(a, 3ull, mask[(m + S) % M]);
                ALL7(F)
#undef F
            } else if (MODE == 1) {
#define F(OFF, S) add32<OFF>(a, 3u, mask[(m + S) % M]);
                ALL7(F)
#undef F
            } else if (MODE == 2) {
#define F(OFF, S) add32<OFF>(a, 3u, mask[(m + S) % M]); add32<OFF + 4>(a, 1u, mask[(m + S) % M]);
                ALL7(F)
#undef F
            } else if (MODE == 3) {
                uint32_t o[7];
#define F(OFF, S) o[S] = add32r<OFF>(a, 3u, mask[(m + S) % M]);
                ALL7(F)
#undef F
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(o[0]), "+v"(o[1]), "+v"(o[2]), "+v"(o[3]), "+v"(o[4]), "+v"(o[5]), "+v"(o[6]));
                acc += (o[0] ^ o[1]) + (o[2] ^ o[3]) + (o[4] ^ o[5]) + o[6];
            } else {
                unsigned long long o[7];
#define F(OFF, S) o[S] = add64r<OFF>(a, 3ull, mask[(m + S) % M]);
                ALL7(F)
#undef F
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(o[0]), "+v"(o[1]), "+v"(o[2]), "+v"(o[3]), "+v"(o[4]), "+v"(o[5]), "+v"(o[6]));
                acc += (uint32_t)((o[0] ^ o[1]) + (o[2] ^ o[3]) + (o[4] ^ o[5]) + o[6]);
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int MODE>
int run(const char* name, int per_iter, int iters, int active, uint32_t* sink) {
    const size_t lds = ROWS * T * 8;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_add<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_add<MODE>), dim3(256), dim3(1024), lds, 0, iters / 10, active, sink);
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((k_add<MODE>), dim3(256), dim3(1024), lds, 0, iters, active, sink);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double wave_instr = 256.0 * 16.0 * (double)iters * per_iter;
    const double rate = wave_instr / (ms * 1e-3);
    printf("%-8s active %2d  %8.3f ms  %7.2f cycles per 49-add group per CU  %.2f cycles per wave-instr per CU @2.4 GHz\n",
           name, active, ms, 2.4e9 * 256.0 / rate * per_iter, 2.4e9 * 256.0 / rate);
    return 0;
}

int main(int argc, char** argv) {
    const int iters = 4000;
    uint32_t* sink;
    CK(hipMalloc(&sink, 4));
    const int actives[] = {1, 8, 15, 32, 64};
    for (int active : actives) {
        if (argc > 1 && atoi(argv[1]) != active) continue;
        if (run<0>("u64", 49, iters, active, sink)) return 1;
        if (run<1>("u32", 49, iters, active, sink)) return 1;
        if (run<2>("u32x2", 98, iters, active, sink)) return 1;
        if (run<3>("rtn32", 49, iters, active, sink)) return 1;
        if (run<4>("rtn64", 49, iters, active, sink)) return 1;
    }
    return 0;
}
