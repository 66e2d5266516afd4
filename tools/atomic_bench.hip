// Fixed cost of folding the per-block count tables of one EM pass:
//   (a) today: every block stores its partial table, k_reduce_partials sums them, k_update consumes
//   (b) every block adds its table into R replicas of one global 64-bit table (no-return device-scope
//       atomics), the consumer sums the R replicas
// 256 blocks x 1024 threads, CELLS = 1283 (K=2, W=20: 1280 counts + 3 statistics).  Times are per pass,
// back to back on one stream, as the iteration loop issues them.
//   hipcc --offload-arch=gfx950 -O3 tools/atomic_bench.hip -o tools/atomic_bench && tools/atomic_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void __launch_bounds__(1024) k_work(unsigned long long* sink, int spin) {
    // stand-in for the sequence kernel's body: a little ALU work so that blocks do not finish in lock step
    unsigned long long x = threadIdx.x + blockIdx.x;
    for (int i = 0; i < spin + (int)(blockIdx.x & 7) * 16; i++) x = x * 6364136223846793005ull + 1442695040888963407ull;
    if (x == 42) sink[0] = x;
}

__global__ void __launch_bounds__(1024) k_atomic(unsigned long long* table, int cells, int replicas, int stride, int spin) {
    unsigned long long x = threadIdx.x + blockIdx.x;
    for (int i = 0; i < spin + (int)(blockIdx.x & 7) * 16; i++) x = x * 6364136223846793005ull + 1442695040888963407ull;
    unsigned long long* t = table + (size_t)(blockIdx.x % replicas) * stride;
    for (int i = threadIdx.x; i < cells; i += blockDim.x)
        (void)__hip_atomic_fetch_add(&t[i], (x & 1023ull) + 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ void __launch_bounds__(1024) k_store(unsigned long long* partial, int cells, int spin) {
    unsigned long long x = threadIdx.x + blockIdx.x;
    for (int i = 0; i < spin + (int)(blockIdx.x & 7) * 16; i++) x = x * 6364136223846793005ull + 1442695040888963407ull;
    unsigned long long* t = partial + (size_t)blockIdx.x * cells;
    for (int i = threadIdx.x; i < cells; i += blockDim.x) t[i] = (x & 1023ull) + 1ull;
}

// today's reduction: 64 cells per block, 16 groups of threads stride over the blocks' partials
__global__ void __launch_bounds__(1024) k_reduce(const unsigned long long* partial, int blocks, int cells, double* red) {
    __shared__ double sh[16][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), g = threadIdx.x >> 6;
    double acc = 0.0;
    if (c < cells) for (int b = g; b < blocks; b += 16) acc += (double)partial[(size_t)b * cells + c];
    sh[g][threadIdx.x & 63] = acc;
    __syncthreads();
    if (g == 0 && c < cells) { double t = 0; for (int i = 0; i < 16; i++) t += sh[i][threadIdx.x]; red[c] = t; }
}

// the consumer (k_update's first step): one block sums the replicas, and clears them for the next pass
__global__ void __launch_bounds__(1024) k_consume_replicas(unsigned long long* table, int cells, int replicas, int stride, float* out) {
    for (int i = threadIdx.x; i < cells; i += blockDim.x) {
        unsigned long long acc = 0;
        for (int r = 0; r < replicas; r++) { acc += table[(size_t)r * stride + i]; table[(size_t)r * stride + i] = 0ull; }
        out[i] = (float)((double)acc * (1.0 / 1099511627776.0));
    }
}
__global__ void __launch_bounds__(1024) k_consume_red(const double* red, int cells, float* out) {
    for (int i = threadIdx.x; i < cells; i += blockDim.x) out[i] = (float)red[i];
}

int main() {
    const int blocks = 256, cells = 1283, stride = 1344 /* 10.5 KiB: replicas on lines of their own */, reps = 300;
    unsigned long long *table, *partial, *sink;
    double* red;
    float* out;
    CK(hipMalloc(&table, (size_t)256 * stride * 8));
    CK(hipMalloc(&partial, (size_t)blocks * cells * 8));
    CK(hipMalloc(&red, cells * 8));
    CK(hipMalloc(&out, cells * 4));
    CK(hipMalloc(&sink, 8));
    CK(hipMemset(table, 0, (size_t)256 * stride * 8));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int spin : {0, 2000, 20000}) {
        // baseline: the body alone + the consumer
        auto time_it = [&](auto&& body, const char* name) -> int {
            for (int i = 0; i < 20; i++) body();
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < reps; i++) body();
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("spin %5d  %-44s %7.2f us per pass\n", spin, name, ms * 1e3f / reps);
            return 0;
        };
        if (time_it([&] { hipLaunchKernelGGL(k_work, dim3(blocks), dim3(1024), 0, st, sink, spin);
                          hipLaunchKernelGGL(k_consume_red, dim3(1), dim3(1024), 0, st, red, cells, out); }, "body + consumer (no folding at all)")) return 1;
        if (time_it([&] { hipLaunchKernelGGL(k_store, dim3(blocks), dim3(1024), 0, st, partial, cells, spin);
                          hipLaunchKernelGGL(k_reduce, dim3((cells + 63) / 64), dim3(1024), 0, st, partial, blocks, cells, red);
                          hipLaunchKernelGGL(k_consume_red, dim3(1), dim3(1024), 0, st, red, cells, out); }, "stores + reduce kernel + consumer (today)")) return 1;
        for (int R : {1, 2, 4, 8, 16, 32, 64, 256}) {
            char name[64];
            snprintf(name, sizeof name, "atomics into %3d replica(s) + consumer", R);
            if (time_it([&] { hipLaunchKernelGGL(k_atomic, dim3(blocks), dim3(1024), 0, st, table, cells, R, stride, spin);
                              hipLaunchKernelGGL(k_consume_replicas, dim3(1), dim3(1024), 0, st, table, cells, R, stride, out); }, name)) return 1;
        }
    }
    // correctness of the atomic path: every cell of `out` after one pass = sum over blocks of the addend
    return 0;
}
