// Rate of scattered no-return device-scope 64-bit atomics (global_atomic_add_x2) from all CUs into one small
// table: what k_em_grp's K = 3 fix lanes do with the counts of their virtual rows.
// 256 blocks x 16 waves; per iteration a wave issues ONE atomic instruction with `active` lanes on random
// cells of a `cells`-cell table, then spins for `spin` multiply-adds (the rest of a sequence's work).
//   hipcc --offload-arch=gfx950 -O3 tools/atomic_scatter_bench.hip -o tools/atomic_scatter_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int SCOPE>
__global__ void __launch_bounds__(1024) k_scatter(unsigned long long* table, int cells, int active, int iters, int spin, float* sink) {
    uint32_t x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    const int lane = threadIdx.x & 63;
    float f = (float)lane;
    for (int it = 0; it < iters; it++) {
        x = x * 1664525u + 1013904223u;
        if (lane < active) {
            unsigned long long* p = table + (x >> 8) % (uint32_t)cells;
            if (SCOPE == 0) (void)__hip_atomic_fetch_add(p, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else (void)__hip_atomic_fetch_add(p, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        for (int i = 0; i < spin; i++) f = f * 1.0000001f + 0.5f;
    }
    if (f == 12345.0f) sink[0] = f;
}

template <int SCOPE>
int run(const char* name, unsigned long long* table, int cells, int active, int iters, int spin, float* sink) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_scatter<SCOPE>, dim3(256), dim3(1024), 0, 0, table, cells, active, iters / 10, spin, sink);
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k_scatter<SCOPE>, dim3(256), dim3(1024), 0, 0, table, cells, active, iters, spin, sink);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double n = 256.0 * 16.0 * iters * active;
    printf("%-9s cells %6d  active %2d  spin %5d  %8.3f ms  %7.2f G lane-atomics/s  %7.1f ns per wave-iteration\n", name, cells, active, spin, ms,
           n / ms * 1e-6, ms * 1e6 / iters);
    return 0;
}

int main() {
    unsigned long long* table; float* sink;
    CK(hipMalloc(&table, 1 << 22)); CK(hipMemset(table, 0, 1 << 22)); CK(hipMalloc(&sink, 4));
    const int iters = 2000;
    for (int spin : {0, 2000}) {
        for (int cells : {5120, 524288}) {
            for (int active : {1, 4, 16, 64}) {
                if (run<0>("agent", table, cells, active, iters, spin, sink)) return 1;
            }
        }
        if (run<1>("workgroup", table, 5120, 16, iters, spin, sink)) return 1;
    }
    // no atomics at all: the spin alone
    if (run<0>("agent", table, 5120, 0, iters, 2000, sink)) return 1;
    return 0;
}
