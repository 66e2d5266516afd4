// The gap between dependent launches on one stream, by launch method: hipLaunchKernelGGL one after the other, and the same
// chain captured into a hipGraph and launched once.  The kernel spins for a fixed number of clock ticks in 256 blocks, so
// (time of N launches) / N - (ticks / 100 MHz) is what a launch boundary costs the stream.
//   hipcc --offload-arch=gfx950 -O3 tools/launch_gap_probe.hip -o tools/launch_gap_probe && tools/launch_gap_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void __launch_bounds__(1024) k_spin(unsigned long long ticks, unsigned* out) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(2);
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] += 1u;          // a dependence from launch to launch
}

// the same with 160 KB of dynamic LDS per block (one block per CU, as the sequence kernels) and a 640-byte argument block
struct Big { unsigned long long w[80]; };
__global__ void __launch_bounds__(1024) k_spin_lds(unsigned long long ticks, unsigned* out, Big big) {
    extern __shared__ unsigned lds[];
    lds[threadIdx.x] = (unsigned)big.w[threadIdx.x % 80u];
    __syncthreads();
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(2);
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] += lds[5];
}

int main() {
    unsigned* d; CK(hipMalloc(&d, 4)); CK(hipMemset(d, 0, 4));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int N = 400;
    for (unsigned long long us : {20ull, 100ull}) {
        const unsigned long long ticks = us * 100ull;
        for (int i = 0; i < 20; i++) hipLaunchKernelGGL(k_spin, dim3(256), dim3(1024), 0, st, ticks, d);
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < N; i++) hipLaunchKernelGGL(k_spin, dim3(256), dim3(1024), 0, st, ticks, d);
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%3llu us kernel, %d stream launches : %.2f us per launch -> %.2f us per boundary\n", us, N, ms * 1e3 / N, ms * 1e3 / N - (double)us);
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < N; i++) hipLaunchKernelGGL(k_spin, dim3(256), dim3(1024), 0, st, ticks, d);
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        CK(hipGraphLaunch(ge, st));
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%3llu us kernel, graph of %d nodes    : %.2f us per launch -> %.2f us per boundary\n", us, N, ms * 1e3 / N, ms * 1e3 / N - (double)us);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
        Big big{};
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_spin_lds), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        for (int i = 0; i < 20; i++) hipLaunchKernelGGL(k_spin_lds, dim3(256), dim3(1024), 160 * 1024, st, ticks, d, big);
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < N; i++) hipLaunchKernelGGL(k_spin_lds, dim3(256), dim3(1024), 160 * 1024, st, ticks, d, big);
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%3llu us kernel, 160 KB LDS + 640 B args   : %.2f us per launch -> %.2f us per boundary\n", us, ms * 1e3 / N, ms * 1e3 / N - (double)us);
    }
    return 0;
}
