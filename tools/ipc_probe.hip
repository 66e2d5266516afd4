// Two PROCESSES, one inbox each in fine-grained device memory, each mapped into the other through hipIpc handles: kernels
// that hand a buffer and a flag to each other with system-scope stores / loads, every wait bounded by the wall clock.
// What the in-kernel all-reduce of libbamm_em (csrc/update_kernel.h: peer_push / peer_wait, csrc/comm.cpp:
// comm_peer_setup) relies on between the ranks of a torch.distributed.run launch -- checked here without RCCL, so that it
// also runs with both processes on ONE device (RCCL refuses two ranks on a device).
//   hipcc --offload-arch=gfx950 -O2 tools/ipc_probe.hip -o tools/ipc_probe && tools/ipc_probe [rounds] [dev_a] [dev_b]
// prints the rounds completed by either side, whether every buffer arrived intact, and microseconds per hop.
#include <hip/hip_runtime.h>
#include <sys/wait.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "[%s] %s: %s\n", who, #x, hipGetErrorString(e_)); return 2; } } while (0)

constexpr unsigned WORDS = 1283, STRIDE = 2056;              // the accumulator of k = 2, W = 20; the library's buffer stride

// side 0 starts; a round: wait for the peer's flag == round (not side 0 in round 1), check its buffer, write ours + flag
__global__ void pingpong(long long* mine, long long* theirs, unsigned side, unsigned rounds, unsigned long long timeout_ticks,
                         unsigned long long* out /* [0] rounds done, [1] bad words, [2] ticks */) {
    __shared__ int fail;
    if (threadIdx.x == 0) fail = 0;
    __syncthreads();
    unsigned long long bad = 0, t_begin = wall_clock64();
    unsigned done = 0;
    for (unsigned r = 1; r <= rounds; r++) {
        const unsigned slot = r % 3u;
        if (!(side == 0 && r == 1)) {                        // wait for the peer's message of round (side 0: r - 1, side 1: r)
            const unsigned want = side == 0 ? r - 1 : r;
            const unsigned wslot = want % 3u;
            if (threadIdx.x == 0) {
                const unsigned long long t0 = wall_clock64();
                while (__hip_atomic_load(mine + (size_t)wslot * STRIDE + STRIDE - 1, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != (long long)want) {
                    if (wall_clock64() - t0 > timeout_ticks) { fail = 1; break; }
                    __builtin_amdgcn_s_sleep(8);
                }
            }
            __syncthreads();
            if (fail) break;
            for (unsigned i = threadIdx.x; i < WORDS; i += blockDim.x) {
                const long long x = __hip_atomic_load(mine + (size_t)wslot * STRIDE + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                if (x != (long long)want * 1000003ll + i) bad++;
            }
        }
        for (unsigned i = threadIdx.x; i < WORDS; i += blockDim.x)
            __hip_atomic_store(theirs + (size_t)slot * STRIDE + i, (long long)r * 1000003ll + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0)
            __hip_atomic_store(theirs + (size_t)slot * STRIDE + STRIDE - 1, (long long)r, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        done = r;
    }
    atomicAdd(out + 1, bad);
    if (threadIdx.x == 0) { out[0] = done; out[2] = wall_clock64() - t_begin; }
}

static int run_side(unsigned side, int dev, unsigned rounds, int rd, int wr) {
    const char* who = side ? "B" : "A";
    CHECK(hipSetDevice(dev));
    long long* inbox = nullptr;
    CHECK(hipExtMallocWithFlags((void**)&inbox, 3 * STRIDE * sizeof(long long), hipDeviceMallocFinegrained));
    CHECK(hipMemset(inbox, 0, 3 * STRIDE * sizeof(long long)));
    hipIpcMemHandle_t h, other;
    CHECK(hipIpcGetMemHandle(&h, inbox));
    if (write(wr, &h, sizeof h) != (ssize_t)sizeof h || read(rd, &other, sizeof other) != (ssize_t)sizeof other) { fprintf(stderr, "[%s] pipe\n", who); return 2; }
    void* peer = nullptr;
    CHECK(hipIpcOpenMemHandle(&peer, other, hipIpcMemLazyEnablePeerAccess));
    unsigned long long* out = nullptr;
    CHECK(hipMalloc((void**)&out, 3 * sizeof(unsigned long long)));
    CHECK(hipMemset(out, 0, 3 * sizeof(unsigned long long)));
    char go = 1;                                             // both sides have mapped: start together
    if (write(wr, &go, 1) != 1 || read(rd, &go, 1) != 1) return 2;
    hipLaunchKernelGGL(pingpong, dim3(1), dim3(1024), 0, 0, inbox, (long long*)peer, side, rounds, 5ull * 100000000ull, out);
    CHECK(hipDeviceSynchronize());
    unsigned long long res[3];
    CHECK(hipMemcpy(res, out, sizeof res, hipMemcpyDeviceToHost));
    printf("[%s] device %d: %llu of %u rounds, %llu bad words, %.2f us per hop (a hop: %u words + flag, stored at system scope, "
           "seen by a kernel of the other process)\n", who, dev, res[0], rounds, res[1], res[0] ? res[2] / 100.0 / (2.0 * res[0]) : 0.0, WORDS);
    if (write(wr, &go, 1) != 1 || read(rd, &go, 1) != 1) return 2;      // nobody unmaps while the other still runs
    CHECK(hipIpcCloseMemHandle(peer));
    CHECK(hipFree(inbox));
    return (res[0] == rounds && res[1] == 0) ? 0 : 1;
}

int main(int argc, char** argv) {
    const unsigned rounds = argc > 1 ? (unsigned)atoi(argv[1]) : 2000u;
    const int dev_a = argc > 2 ? atoi(argv[2]) : 0, dev_b = argc > 3 ? atoi(argv[3]) : 0;
    int ab[2], ba[2];
    if (pipe(ab) || pipe(ba)) return 2;
    const pid_t child = fork();                              // before anything touches the GPU
    if (child == 0) return run_side(1, dev_b, rounds, ab[0], ba[1]);
    const int rc = run_side(0, dev_a, rounds, ba[0], ab[1]);
    int st = 0;
    waitpid(child, &st, 0);
    const int rcb = WIFEXITED(st) ? WEXITSTATUS(st) : 3;
    printf("ipc_probe: %s\n", (rc == 0 && rcb == 0) ? "OK" : "FAILED");
    return rc ? rc : rcb;
}
