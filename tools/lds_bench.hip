// LDS micro-benchmarks that size the E-step gather and the M-step histogram on gfx950.
// Build: hipcc --offload-arch=gfx950 -O3 tools/lds_bench.hip -o tools/lds_bench
// Each kernel runs ITER rounds of 8 LDS wave-instructions per wave; the address pattern is what
// differs.  Output: ns per wave-instruction per CU (all resident waves issuing), and the lane rate.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int ITER = 2000;

__device__ __forceinline__ unsigned lcg(unsigned& s) { s = s * 1664525u + 1013904223u; return s >> 8; }
// cheap per-op address update (2 VALU): keeps the LDS pipe, not the VALU, the bottleneck
#define NEXT(a) ((a) * 13u + 7u)

enum Pattern { LANE = 0, RAND64 = 1, RAND1280 = 2, SAME = 3, RAND64_PRIV16 = 4, RAND320 = 5, RAND64_PRIV32 = 6, RAND64_PRIV8 = 7 };

template <int PAT>
__device__ __forceinline__ unsigned addr_of(unsigned r, int lane) {
    if (PAT == LANE) return lane + (r & 7u) * 64u;                       // conflict-free
    if (PAT == RAND64) return (r & 63u) + ((r >> 6) & 7u) * 64u * 0u;     // 64 rows of one column
    if (PAT == RAND1280) return r % 1280u;
    if (PAT == SAME) return 5u;
    if (PAT == RAND64_PRIV16) return (r & 63u) * 16u + (lane & 15u);      // 16 copies, copy = lane%16
    if (PAT == RAND64_PRIV32) return (r & 63u) * 32u + (lane & 31u);      // 32 copies: bank = lane%32
    if (PAT == RAND64_PRIV8) return (r & 63u) * 8u + (lane & 7u);
    if (PAT == RAND320) return r % 320u;
    return 0;
}

template <int PAT>
__global__ void __launch_bounds__(1024) k_read(float* out) {
    __shared__ float lds[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = 1.0f + i * 1e-6f;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    unsigned s = threadIdx.x * 2654435761u + blockIdx.x;
    float acc = 0.f;
    for (int it = 0; it < ITER; it++) {
        unsigned a[8];
#pragma unroll
        for (int u = 0; u < 8; u++) a[u] = addr_of<PAT>(lcg(s), lane);
#pragma unroll
        for (int u = 0; u < 8; u++) acc += lds[a[u]];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int PAT>
__global__ void __launch_bounds__(1024) k_read64(float* out) {
    __shared__ float2 lds[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = make_float2(1.0f + i * 1e-6f, 0.5f);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    unsigned s = threadIdx.x * 2654435761u + blockIdx.x;
    float acc = 0.f;
    for (int it = 0; it < ITER; it++) {
        unsigned a[8];
#pragma unroll
        for (int u = 0; u < 8; u++) a[u] = addr_of<PAT>(lcg(s), lane);
#pragma unroll
        for (int u = 0; u < 8; u++) { float2 v = lds[a[u]]; acc += v.x * v.y; }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int PAT>
__global__ void __launch_bounds__(1024) k_atomic_f32(float* out) {
    __shared__ float lds[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    unsigned s = threadIdx.x * 2654435761u + blockIdx.x;
    for (int it = 0; it < ITER; it++) {
        unsigned a[8];
#pragma unroll
        for (int u = 0; u < 8; u++) a[u] = addr_of<PAT>(lcg(s), lane);
#pragma unroll
        for (int u = 0; u < 8; u++) atomicAdd(&lds[a[u]], 1.0f);
    }
    __syncthreads();
    out[blockIdx.x * blockDim.x + threadIdx.x] = lds[threadIdx.x];
}

template <int PAT>
__global__ void __launch_bounds__(1024) k_atomic_u32(float* out) {
    __shared__ unsigned lds[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    unsigned s = threadIdx.x * 2654435761u + blockIdx.x;
    for (int it = 0; it < ITER; it++) {
        unsigned a[8];
#pragma unroll
        for (int u = 0; u < 8; u++) a[u] = addr_of<PAT>(lcg(s), lane);
#pragma unroll
        for (int u = 0; u < 8; u++) atomicAdd(&lds[a[u]], 1u);
    }
    __syncthreads();
    out[blockIdx.x * blockDim.x + threadIdx.x] = (float)lds[threadIdx.x];
}

// non-atomic read-modify-write (only correct when no two lanes/waves share an address)
template <int PAT>
__global__ void __launch_bounds__(1024) k_rmw(float* out) {
    __shared__ float lds[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    unsigned s = threadIdx.x * 2654435761u + blockIdx.x;
    for (int it = 0; it < ITER; it++) {
        unsigned a[8];
#pragma unroll
        for (int u = 0; u < 8; u++) a[u] = addr_of<PAT>(lcg(s), lane);
#pragma unroll
        for (int u = 0; u < 8; u++) { float v = lds[a[u]]; lds[a[u]] = v + 1.0f; }
    }
    __syncthreads();
    out[blockIdx.x * blockDim.x + threadIdx.x] = lds[threadIdx.x];
}

template <int PAT>
__global__ void __launch_bounds__(1024) k_atomic_u64(float* out) {
    __shared__ unsigned long long lds[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    unsigned s = threadIdx.x * 2654435761u + blockIdx.x;
    for (int it = 0; it < ITER; it++) {
        unsigned a[8];
#pragma unroll
        for (int u = 0; u < 8; u++) a[u] = addr_of<PAT>(lcg(s), lane);
#pragma unroll
        for (int u = 0; u < 8; u++) atomicAdd(&lds[a[u]], (unsigned long long)(a[u] + 1));
    }
    __syncthreads();
    out[blockIdx.x * blockDim.x + threadIdx.x] = (float)lds[threadIdx.x];
}

template <int PAT>
__global__ void __launch_bounds__(1024) k_atomic_f64(float* out) {
    __shared__ double lds[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    unsigned s = threadIdx.x * 2654435761u + blockIdx.x;
    for (int it = 0; it < ITER; it++) {
        unsigned a[8];
#pragma unroll
        for (int u = 0; u < 8; u++) a[u] = addr_of<PAT>(lcg(s), lane);
#pragma unroll
        for (int u = 0; u < 8; u++) atomicAdd(&lds[a[u]], 1.0);
    }
    __syncthreads();
    out[blockIdx.x * blockDim.x + threadIdx.x] = (float)lds[threadIdx.x];
}

// two u32 atomics per value (hi/lo limbs)
template <int PAT>
__global__ void __launch_bounds__(1024) k_atomic_2xu32(float* out) {
    __shared__ unsigned lds[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    unsigned s = threadIdx.x * 2654435761u + blockIdx.x;
    for (int it = 0; it < ITER; it++) {
        unsigned a[8];
#pragma unroll
        for (int u = 0; u < 8; u++) a[u] = addr_of<PAT>(lcg(s), lane);
#pragma unroll
        for (int u = 0; u < 8; u++) { atomicAdd(&lds[a[u]], a[u]); atomicAdd(&lds[a[u] + 2048], 1u); }
    }
    __syncthreads();
    out[blockIdx.x * blockDim.x + threadIdx.x] = (float)lds[threadIdx.x];
}

__global__ void __launch_bounds__(1024) k_bpermute(float* out) {
    const int lane = threadIdx.x & 63;
    unsigned s = threadIdx.x * 2654435761u + blockIdx.x;
    float acc = 0.f, x = (float)lane;
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) acc += __shfl(x, lcg(s) & 63u, 64);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

__global__ void __launch_bounds__(1024) k_valu_only(float* out) {
    const int lane = threadIdx.x & 63;
    unsigned s = threadIdx.x * 2654435761u + blockIdx.x;
    float acc = 0.f;
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) acc += (float)(addr_of<RAND64>(lcg(s), lane));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <class F>
void run(const char* name, F launch, int blocks, int threads, float* d_out, int cus) {
    launch(blocks, threads, d_out);
    CHECK(hipDeviceSynchronize());
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    CHECK(hipEventRecord(a));
    for (int r = 0; r < 3; r++) launch(blocks, threads, d_out);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms;
    CHECK(hipEventElapsedTime(&ms, a, b));
    ms /= 3;
    const double wave_instr = (double)blocks * (threads / 64) * ITER * 8;
    const double per_cu = wave_instr / cus;
    printf("%-28s blocks=%4d thr=%4d  %8.3f ms  %7.2f ns/wave-instr/CU  %8.2f Glane/s\n", name, blocks, threads, ms,
           ms * 1e6 / per_cu, wave_instr * 64 / (ms * 1e-3) / 1e9);
}

int main(int argc, char** argv) {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("device %s, %d CUs, clock %d kHz\n", prop.name, cus, prop.clockRate);
    float* d_out;
    CHECK(hipMalloc(&d_out, 4096 * 1024 * sizeof(float)));
#define RUN(NAME, KERNEL, B, T) run(NAME, [](int b, int t, float* o) { hipLaunchKernelGGL(KERNEL, dim3(b), dim3(t), 0, 0, o); }, B, T, d_out, cus)
    for (int mult = 2; mult <= 2; mult++) {
        const int B = cus * mult, T = 1024;
        printf("---- %d blocks/CU x %d threads\n", mult, T);
        RUN("valu only (address gen)", k_valu_only, B, T);
        RUN("read_b32 lane", k_read<LANE>, B, T);
        RUN("read_b32 rand64", k_read<RAND64>, B, T);
        RUN("read_b32 rand320", k_read<RAND320>, B, T);
        RUN("read_b32 rand1280", k_read<RAND1280>, B, T);
        RUN("read_b32 same", k_read<SAME>, B, T);
        RUN("read_b64 lane", k_read64<LANE>, B, T);
        RUN("read_b64 rand64", k_read64<RAND64>, B, T);
        RUN("read_b64 rand1280", k_read64<RAND1280>, B, T);
        RUN("bpermute rand", k_bpermute, B, T);
        RUN("atomic_f32 lane", k_atomic_f32<LANE>, B, T);
        RUN("atomic_f32 rand64", k_atomic_f32<RAND64>, B, T);
        RUN("atomic_f32 rand320", k_atomic_f32<RAND320>, B, T);
        RUN("atomic_f32 rand1280", k_atomic_f32<RAND1280>, B, T);
        RUN("atomic_f32 same", k_atomic_f32<SAME>, B, T);
        RUN("atomic_f32 rand64 priv8", k_atomic_f32<RAND64_PRIV8>, B, T);
        RUN("atomic_f32 rand64 priv16", k_atomic_f32<RAND64_PRIV16>, B, T);
        RUN("atomic_f32 rand64 priv32", k_atomic_f32<RAND64_PRIV32>, B, T);
        RUN("atomic_u64 lane", k_atomic_u64<LANE>, B, T);
        RUN("atomic_u64 rand64", k_atomic_u64<RAND64>, B, T);
        RUN("atomic_u64 rand1280", k_atomic_u64<RAND1280>, B, T);
        RUN("atomic_f64 lane", k_atomic_f64<LANE>, B, T);
        RUN("atomic_f64 rand64", k_atomic_f64<RAND64>, B, T);
        RUN("atomic_2xu32 rand64 (per pair)", k_atomic_2xu32<RAND64>, B, T);
        RUN("atomic_2xu32 rand1280 (per pair)", k_atomic_2xu32<RAND1280>, B, T);
        RUN("atomic_u32 lane", k_atomic_u32<LANE>, B, T);
        RUN("atomic_u32 rand64", k_atomic_u32<RAND64>, B, T);
        RUN("atomic_u32 rand1280", k_atomic_u32<RAND1280>, B, T);
        RUN("rmw lane", k_rmw<LANE>, B, T);
        RUN("rmw rand64 priv32", k_rmw<RAND64_PRIV32>, B, T);
    }
    return 0;
}
