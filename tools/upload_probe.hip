// What a hipMemcpyAsync from PAGEABLE host memory costs on the host side, by how the source pages came to be
// (bamm_em_create's upload of the per-order exception offsets took 20-22 ms for 8 MB in every trace of round 5 while its
// neighbours of the same size took 0.3 ms).   hipcc --offload-arch=gfx950 -O2 -o upload_probe upload_probe.hip -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    const size_t n = (size_t)1000001;
    hipStream_t st;
    hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    uint64_t* d = nullptr;
    hipMalloc(&d, n * 8);
    auto up = [&](const char* what, const void* src, size_t bytes) {
        const double t = now();
        hipMemcpyAsync(d, src, bytes, hipMemcpyHostToDevice, st);
        const double t1 = now();
        hipStreamSynchronize(st);
        printf("%8.3f ms call + %6.3f ms sync   %s\n", t1 - t, now() - t1, what);
    };
    {   // a big first copy, as the resident set's stream is
        std::vector<uint32_t> big(26000000, 1u);
        uint32_t* dbig = nullptr; hipMalloc(&dbig, big.size() * 4);
        const double t = now();
        hipMemcpyAsync(dbig, big.data(), big.size() * 4, hipMemcpyHostToDevice, st); hipStreamSynchronize(st);
        printf("%8.3f ms   first copy of the process, 104 MB\n", now() - t);
        hipFree(dbig);
    }
    for (int rep = 0; rep < 2; rep++) {
        { std::vector<uint64_t> v(n); for (size_t i = 0; i < n; i++) v[i] = i; up("vector(n), filled by the main thread", v.data(), n * 8); }
        { std::vector<uint64_t> v; v.assign(n, 0); up("assign(n, 0) only", v.data(), n * 8); }
        {
            std::vector<uint64_t> v; v.assign(n, 0);
            std::vector<std::thread> th;
            for (int t = 0; t < 16; t++) th.emplace_back([&v, t, n] { for (size_t i = n * t / 16; i < n * (t + 1) / 16; i++) v[i] = i; });
            for (auto& x : th) x.join();
            up("assign(n, 0), then filled by 16 threads", v.data(), n * 8);
        }
        {
            std::vector<uint64_t> v; v.assign(n, 0);
            std::vector<std::thread> th;
            for (int t = 0; t < 16; t++) th.emplace_back([&v, t, n] { for (size_t i = n * t / 16; i < n * (t + 1) / 16; i++) v[i] = i; });
            for (auto& x : th) x.join();
            for (size_t i = 0; i + 1 < n; i++) v[i + 1] += v[i];
            up("the same + a serial prefix sum by the main thread", v.data(), n * 8);
        }
        {
            uint64_t* p = nullptr; hipHostMalloc((void**)&p, n * 8, hipHostMallocDefault);
            std::vector<uint64_t> v(n, 3);
            const double t = now(); memcpy(p, v.data(), n * 8); const double t1 = now();
            up("from pinned memory (after a memcpy into it, see next line)", p, n * 8);
            printf("%8.3f ms   that memcpy\n", t1 - t);
            hipHostFree(p);
        }
    }
    // device -> pageable host, then a kernel launch: does the launch wait?
    {
        std::vector<float> out(800000);
        float* dsrc = nullptr; hipMalloc(&dsrc, out.size() * 4);
        for (int rep = 0; rep < 3; rep++) {
            double t = now();
            hipMemcpyAsync(out.data(), dsrc, out.size() * 4, hipMemcpyDeviceToHost, st); hipStreamSynchronize(st);
            const double t1 = now();
            hipMemsetAsync(d, 0, 64, st); hipStreamSynchronize(st);
            printf("%8.3f ms D2H of 3.2 MB into pageable memory, then %6.3f ms for a memset + sync behind it\n", t1 - t, now() - t1);
        }
        float* tmp = nullptr; hipMalloc(&tmp, 3200000);
        double t = now(); hipFree(tmp); double t1 = now();
        hipMemsetAsync(d, 0, 64, st); hipStreamSynchronize(st);
        printf("%8.3f ms hipFree of 3.2 MB, then %6.3f ms for a memset + sync behind it\n", t1 - t, now() - t1);
    }
    return 0;
}
