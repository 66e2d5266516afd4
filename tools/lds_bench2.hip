// LDS micro-benchmarks, round 2: candidate primitives for the E-step gather and the M-step
// histogram with cheap address streams (2 full-rate VALU per op) so that the LDS pipe is the
// bottleneck.  Rows are drawn from a 64-row column (k=2) unless stated.
// Build: hipcc --offload-arch=gfx950 -O3 tools/lds_bench2.hip -o tools/lds_bench2
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int ITER = 4000;
constexpr int UNR = 8;

// per-lane row stream: r[u] <- (r[u] + odd) & 63 : 2 full-rate VALU, rows stay "random-like"
#define INIT_ROWS                                                       \
    unsigned r[UNR];                                                    \
    {                                                                   \
        unsigned s = (threadIdx.x + 1u) * 2654435761u + blockIdx.x * 40503u; \
        for (int u = 0; u < UNR; u++) { s = s * 1664525u + 1013904223u; r[u] = (s >> 10) & 63u; } \
    }                                                                   \
    const unsigned inc = ((threadIdx.x * 7u + blockIdx.x) | 1u) & 63u;
#define STEP_ROW(u) r[u] = (r[u] + inc) & 63u
// 320-row variant (pair tables): add an odd step, wrap by conditional subtract (3 VALU)
#define STEP_ROW320(u) { r[u] += inc320; r[u] = r[u] >= 320u ? r[u] - 320u : r[u]; }

template <int WORDS>  // 1: b32, 2: b64, 4: b128 per row
__global__ void __launch_bounds__(1024) k_read(float* out, int rows) {
    extern __shared__ __align__(16) float lds[];   // without the alignment hipcc splits the wide reads
    for (int i = threadIdx.x; i < rows * WORDS; i += blockDim.x) lds[i] = 1.0f;
    __syncthreads();
    INIT_ROWS
    float acc = 0.f;
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int u = 0; u < UNR; u++) {
            STEP_ROW(u);
            if (WORDS == 1) acc += lds[r[u]];
            if (WORDS == 2) { float2 v = reinterpret_cast<float2*>(lds)[r[u]]; acc += v.x + v.y; }
            if (WORDS == 4) { float4 v = reinterpret_cast<float4*>(lds)[r[u]]; acc += v.x + v.y + v.z + v.w; }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

// C private copies of the column, copy = lane % C, layout [row][C]
template <int C, int MODE>  // MODE 0: u32 no-return, 1: u32 with return (carry detect), 2: u64, 3: f64
__global__ void __launch_bounds__(1024) k_atomic(float* out) {
    extern __shared__ unsigned lds32[];
    unsigned long long* lds64 = reinterpret_cast<unsigned long long*>(lds32);
    double* ldsd = reinterpret_cast<double*>(lds32);
    for (int i = threadIdx.x; i < 64 * C * 2 + 64; i += blockDim.x) lds32[i] = 0;
    __syncthreads();
    INIT_ROWS
    const unsigned copy = (threadIdx.x & 63u) % C;
    unsigned carries = 0;
    const unsigned x = 0x01000000u + threadIdx.x;
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int u = 0; u < UNR; u++) {
            STEP_ROW(u);
            const unsigned a = r[u] * C + copy;
            if (MODE == 0) atomicAdd(&lds32[a], x);
            if (MODE == 1) {
                const unsigned old = atomicAdd(&lds32[a], x);
                if (old + x < old) atomicAdd(&lds32[64 * C + r[u]], 1u);   // rare carry into the high word
            }
            if (MODE == 2) atomicAdd(&lds64[a], (unsigned long long)x << 8);
            if (MODE == 3) atomicAdd(&ldsd[a], 1.0);
        }
    }
    __syncthreads();
    out[blockIdx.x * blockDim.x + threadIdx.x] = (float)lds32[threadIdx.x] + carries;
}

// b64 gather from a [64 rows (+1 pad)][2 floats] column pair, optionally two stages of b32 instead
template <int MODE>
__global__ void __launch_bounds__(1024) k_pairread(float* out) {
    extern __shared__ __align__(16) float lds[];
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) lds[i] = 1.0f;
    __syncthreads();
    INIT_ROWS
    float acc = 0.f;
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int u = 0; u < UNR; u++) {
            STEP_ROW(u);
            if (MODE == 0) { float2 v = reinterpret_cast<float2*>(lds)[r[u]]; acc = acc * v.x + v.y; }
            if (MODE == 1) { acc = acc * lds[r[u]] + lds[65 + r[u]]; }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int C, int MODE>  // MODE 0: b32 read, 2: u64 atomic ; 320 rows
__global__ void __launch_bounds__(1024) k_rows320(float* out) {
    extern __shared__ unsigned lds32[];
    unsigned long long* lds64 = reinterpret_cast<unsigned long long*>(lds32);
    float* ldsf = reinterpret_cast<float*>(lds32);
    for (int i = threadIdx.x; i < 320 * C * 2; i += blockDim.x) lds32[i] = 0;
    __syncthreads();
    INIT_ROWS
    for (int u = 0; u < UNR; u++) r[u] = (r[u] * 5u + threadIdx.x) % 320u;
    const unsigned inc320 = 1u + 2u * ((threadIdx.x * 7u + blockIdx.x) % 150u);
    const unsigned copy = (threadIdx.x & 63u) % C;
    float acc = 0.f;
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int u = 0; u < UNR; u++) {
            STEP_ROW320(u);
            const unsigned a = r[u] * C + copy;
            if (MODE == 0) acc += ldsf[a];
            if (MODE == 2) atomicAdd(&lds64[a], (unsigned long long)(a + 1) << 8);
        }
    }
    __syncthreads();
    out[blockIdx.x * blockDim.x + threadIdx.x] = (float)lds32[threadIdx.x] + acc + (float)inc;
}

__global__ void __launch_bounds__(1024) k_valu(float* out) {
    INIT_ROWS
    unsigned acc = 0;
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int u = 0; u < UNR; u++) { STEP_ROW(u); acc += r[u]; }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (float)acc;
}

template <class F>
void run(const char* name, F launch, int blocks, int threads, int cus) {
    launch();
    CHECK(hipDeviceSynchronize());
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    CHECK(hipEventRecord(a));
    for (int rr = 0; rr < 3; rr++) launch();
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms;
    CHECK(hipEventElapsedTime(&ms, a, b));
    ms /= 3;
    const double wave_instr = (double)blocks * (threads / 64) * ITER * UNR;
    printf("%-34s %8.3f ms  %6.2f ns/wave-instr/CU  (%5.1f cyc @2.4GHz)\n", name, ms, ms * 1e6 / (wave_instr / cus),
           ms * 1e6 / (wave_instr / cus) * 2.4);
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    float* d_out;
    CHECK(hipMalloc(&d_out, 4096 * 1024 * sizeof(float)));
    for (int wpc = 16; wpc <= 16; wpc += 16) {
        const int B = cus * (wpc / 16), T = 1024;
        printf("---- %d waves per CU\n", wpc);
#define RUNK(NAME, KERNEL, LDS, ...) run(NAME, [&]() { hipLaunchKernelGGL(KERNEL, dim3(B), dim3(T), LDS, 0, __VA_ARGS__); }, B, T, cus)
        RUNK("valu only", k_valu, 0, d_out);
        RUNK("read b32  rand64", k_read<1>, 4096, d_out, 64);
        RUNK("read b64  rand64", k_read<2>, 4096, d_out, 64);
        RUNK("read b128 rand64", k_read<4>, 4096, d_out, 64);
        RUNK("atomic u32        C=1", (k_atomic<1, 0>), 16384, d_out);
        RUNK("atomic u32        C=2", (k_atomic<2, 0>), 16384, d_out);
        RUNK("atomic u32        C=4", (k_atomic<4, 0>), 16384, d_out);
        RUNK("atomic u32        C=8", (k_atomic<8, 0>), 16384, d_out);
        RUNK("atomic u32        C=16", (k_atomic<16, 0>), 16384, d_out);
        RUNK("atomic u32 rtn+carry C=1", (k_atomic<1, 1>), 16384, d_out);
        RUNK("atomic u32 rtn+carry C=4", (k_atomic<4, 1>), 16384, d_out);
        RUNK("atomic u32 rtn+carry C=8", (k_atomic<8, 1>), 16384, d_out);
        RUNK("atomic u64        C=1", (k_atomic<1, 2>), 16384, d_out);
        RUNK("atomic u64        C=2", (k_atomic<2, 2>), 16384, d_out);
        RUNK("atomic u64        C=4", (k_atomic<4, 2>), 16384, d_out);
        RUNK("atomic u64        C=8", (k_atomic<8, 2>), 16384, d_out);
        RUNK("atomic u64        C=16", (k_atomic<16, 2>), 16384, d_out);
        RUNK("pair: 1 x b64 (2 columns)", (k_pairread<0>), 8192, d_out);
        RUNK("pair: 2 x b32 (2 columns)", (k_pairread<1>), 8192, d_out);
        RUNK("read b32  rand320", (k_rows320<1, 0>), 320 * 8, d_out);
        RUNK("atomic u64 rand320 C=1", (k_rows320<1, 2>), 320 * 8, d_out);
        RUNK("atomic u64 rand320 C=2", (k_rows320<2, 2>), 320 * 16, d_out);
        RUNK("atomic u64 rand320 C=4", (k_rows320<4, 2>), 320 * 32, d_out);
        RUNK("atomic f64        C=1", (k_atomic<1, 3>), 16384, d_out);
        RUNK("atomic f64        C=8", (k_atomic<8, 3>), 16384, d_out);
    }
    return 0;
}
