// What the HIP runtime costs a short command at both ends: first use (hipInit, the device's primary context, a stream,
// the first allocation, the first launch of a kernel of this file) and the time between the program's last line and the
// process being reaped -- with the device memory freed first, and left to the driver (`leave`).
//   hipcc --offload-arch=gfx950 -O3 tools/init_probe.hip -o tools/init_probe && python3 tools/init_probe.py
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <unistd.h>

__global__ void k_touch(float* p) { p[threadIdx.x] = 1.0f; }

static double now() { return std::chrono::duration<double>(std::chrono::system_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
    const bool leave = argc > 1 && !strcmp(argv[1], "leave");
    const size_t mb = argc > 2 ? (size_t)atol(argv[2]) : 2048;
    double t = now();
    printf("entered %.4f\n", t);
    auto lap = [&](const char* what) { const double n = now(); printf("  %-44s %.4f s\n", what, n - t); t = n; };
    if (hipInit(0) != hipSuccess) { printf("no device\n"); return 1; }
    lap("hipInit");
    (void)hipSetDevice(0); (void)hipFree(nullptr);
    lap("hipSetDevice + hipFree(0)");
    hipStream_t st; (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    lap("hipStreamCreate");
    float* d = nullptr; (void)hipMalloc(&d, mb << 20);
    lap("hipMalloc");
    void* h = nullptr; (void)hipHostMalloc(&h, 256u << 20, hipHostMallocDefault);
    lap("hipHostMalloc 256 MB");
    hipLaunchKernelGGL(k_touch, dim3(1), dim3(64), 0, st, d);
    (void)hipStreamSynchronize(st);
    lap("first launch + sync");
    (void)hipMemsetAsync(d, 0, mb << 20, st); (void)hipStreamSynchronize(st);
    lap("memset of the allocation");
    const bool reset = argc > 1 && !strcmp(argv[1], "reset");
    if (reset) { (void)hipDeviceReset(); lap("hipDeviceReset"); }
    else if (argc > 1 && !strcmp(argv[1], "free_dev")) { (void)hipFree(d); lap("hipFree"); }
    else if (argc > 1 && !strcmp(argv[1], "free_host")) { (void)hipHostFree(h); lap("hipHostFree"); }
    else if (argc > 1 && !strcmp(argv[1], "free_stream")) { (void)hipStreamDestroy(st); lap("hipStreamDestroy"); }
    else if (!leave) { (void)hipFree(d); (void)hipHostFree(h); (void)hipStreamDestroy(st); lap("hipFree + hipHostFree + stream"); }
    printf("left %.4f\n", now());
    fflush(nullptr);
    _exit(0);
}
