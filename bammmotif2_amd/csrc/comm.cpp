// RCCL inside the drop-in: the one collective of the EM path -- an int64 sum of the fused accumulator
// [n_K | llh | sum_r | n_seqs] per pass -- issued on the context's HIP stream.
//
// Reference: the cross-sequence reductions this stands for are the OpenMP `reduction(+:llikelihood)`
// (/root/reference/src/refinement/EM.cpp:148), the CAS float adds into n_[K] (EM.cpp:203-215,240) and
// the serial sum over r_ of optimize_q (EM.cpp:509-513); sequences shard over the GPUs.
//
// librccl is opened at the first bamm_comm_* call (dlopen + dlsym), not linked: a process that never
// shards loads nothing, and a process that already carries an RCCL (torch ships its own) keeps using that
// copy -- the loader hands back the library that is already mapped under the same SONAME.
#include <dlfcn.h>
#include <fcntl.h>
#include <rccl/rccl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <thread>

#include <atomic>
#include <cstring>
#include <condition_variable>
#include <mutex>

#include "common.h"

using namespace bamm;

// bamm_comm_init_local: the ranks of ONE process summed through pinned host memory -- no RCCL, any devices (even
// one device behind several contexts).  A self-test vehicle (the N > 1 host logic on a 1-GPU box) and the way out
// when librccl cannot be opened; tens of microseconds per call, never used for reported numbers.
struct LocalGroup {
    std::mutex mu;
    std::condition_variable cv;
    uint32_t n = 0, arrived = 0, generation = 0, refs = 0;
    bool aborted = false;
    std::vector<long long*> inbox;          // [rank] pinned, `cap` words: what the rank contributed
    size_t cap = 0;
    // false when the group was aborted while (or before) waiting
    bool barrier() {
        std::unique_lock<std::mutex> lock(mu);
        if (aborted) return false;
        const uint32_t gen = generation;
        if (++arrived == n) { arrived = 0; generation++; cv.notify_all(); return true; }
        cv.wait(lock, [&] { return generation != gen || aborted; });
        return generation != gen;
    }
};

// bamm_comm_init_shm: the same host-staged sum between PROCESSES of one host, through a POSIX shared-memory segment -- what
// bamm_comm_init_local is to the threads of one process.  A self-test vehicle as well: it lets the cross-process half of the
// in-kernel all-reduce (inboxes mapped through hipIpc*MemHandle) run between two processes on a 1-GPU box, where RCCL
// refuses two ranks on one device.  Every wait is bounded.
struct ShmGroup {
    std::atomic<uint32_t> magic;            // kShmMagic once the creator has initialised the header
    std::atomic<uint32_t> arrived, generation, aborted, attached;
    uint32_t n;
    uint64_t cap;
    // the join handshake (bamm_comm_init_shm): rank r writes a fresh nonce into hello[r], the creator answers ack[r] = hello[r]
    std::atomic<unsigned long long> hello[64], ack[64];
    // long long inbox[n][cap] follows
    long long* inbox(uint32_t r) { return reinterpret_cast<long long*>(this + 1) + (size_t)r * cap; }
    // false when the group was aborted or a peer did not arrive within the limit
    bool barrier(double limit_s = 60.0) {
        if (aborted.load(std::memory_order_acquire)) return false;
        const uint32_t gen = generation.load(std::memory_order_acquire);
        if (arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == n) {
            arrived.store(0, std::memory_order_relaxed);
            generation.fetch_add(1, std::memory_order_acq_rel);
            return true;
        }
        const auto t0 = std::chrono::steady_clock::now();
        uint32_t spins = 0;
        while (generation.load(std::memory_order_acquire) == gen) {
            if (aborted.load(std::memory_order_acquire)) return false;
            if (++spins > 2000) std::this_thread::sleep_for(std::chrono::microseconds(50));
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit_s) {
                aborted.store(1, std::memory_order_release);
                return false;
            }
        }
        return true;
    }
};
constexpr uint32_t kShmMagic = 0x42414d4du;
constexpr uint32_t kShmMaxWorld = 64;

struct bamm_comm {
    ShmGroup* shm = nullptr;                // non-null: host-staged sum between processes (bamm_comm_init_shm)
    size_t shm_bytes = 0;
    std::string shm_name;
    // `mu` orders every use of `comm` against bamm_comm_abort / bamm_comm_destroy: ncclCommAbort frees the communicator,
    // so exactly one thread may call it, and nobody may enqueue on the handle afterwards (failing ranks abort ALL the
    // communicators of their group, from several threads at once).  An enqueue never waits for a peer -- the kernel
    // does -- so an aborting thread waits for the lock at most as long as one ncclAllReduce call takes to return.
    std::mutex mu;
    std::atomic<bool> aborted{false};       // set first by bamm_comm_abort: later calls fail with BAMM_ERR_COMM
    ncclComm_t comm = nullptr;
    bamm_ctx* ctx = nullptr;
    uint32_t rank = 0, world = 1;
    LocalGroup* local = nullptr;            // non-null: host-staged sum inside this process
    long long* h_sum = nullptr;             // pinned, local kind
    // in-kernel all-reduce (common.h: PeerArgs): this rank's inbox and the peers' as mapped on this device
    bool peer_tried = false, peer_ready = false;
    uint32_t peer_stride = 0;
    void* peer_inbox = nullptr;
    void* peer_map[kPeerMaxWorld] = {};
    bool peer_ipc[kPeerMaxWorld] = {};      // opened with hipIpcOpenMemHandle (another process's memory)
    unsigned long long peer_seq = 0;        // sequence number of the last pushing pass (the ranks count in step)
    std::string peer_why;                   // why the inboxes could not be set up
};

namespace {

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*GetVersion)(int*) = nullptr;
};

Rccl g_rccl;
std::once_flag g_once;
std::string g_load_error;

void load_rccl() {
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    void* h = nullptr;
    for (const char* n : names)
        if ((h = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
    if (!h) {
        const char* e = dlerror();
        g_load_error = std::string("librccl could not be opened: ") + (e ? e : "not found");
        return;
    }
    Rccl r;
    r.handle = h;
    bool ok = true;
    auto sym = [&](const char* name) { void* p = dlsym(h, name); if (!p) { ok = false; g_load_error = std::string("librccl lacks ") + name; } return p; };
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
    r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(sym("ncclCommInitAll"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
    r.CommAbort = reinterpret_cast<decltype(r.CommAbort)>(sym("ncclCommAbort"));
    r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(sym("ncclAllReduce"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    r.GetVersion = reinterpret_cast<decltype(r.GetVersion)>(sym("ncclGetVersion"));
    if (ok) g_rccl = r;
}

const Rccl* rccl() {
    std::call_once(g_once, load_rccl);
    if (!g_rccl.handle) {
        set_error("%s", g_load_error.c_str());
        return nullptr;
    }
    return &g_rccl;
}

int fail(const Rccl* r, const char* what, ncclResult_t rc) {
    set_error("%s failed: %s", what, r->GetErrorString(rc));
    return BAMM_ERR_COMM;
}

}  // namespace

namespace bamm {

static int local_allreduce_i64(bamm_comm* c, void* dev_ptr, size_t n_words, hipStream_t st) {
    LocalGroup* g = c->local;
    if (n_words > g->cap) { set_error("local all-reduce of %zu words (capacity %zu)", n_words, g->cap); return BAMM_ERR_ARG; }
    BAMM_HIP(hipSetDevice(ctx_device(c->ctx)));
    BAMM_HIP(hipMemcpyAsync(g->inbox[c->rank], dev_ptr, n_words * sizeof(long long), hipMemcpyDeviceToHost, st));
    BAMM_HIP(hipStreamSynchronize(st));
    if (!g->barrier()) { set_error("local all-reduce: the group was aborted"); return BAMM_ERR_COMM; }
    for (size_t i = 0; i < n_words; i++) {                   // integer sums: the rank order does not matter
        long long t = 0;
        for (uint32_t r = 0; r < g->n; r++) t += g->inbox[r][i];
        c->h_sum[i] = t;
    }
    if (!g->barrier()) { set_error("local all-reduce: the group was aborted"); return BAMM_ERR_COMM; }   // inboxes free again
    BAMM_HIP(hipMemcpyAsync(dev_ptr, c->h_sum, n_words * sizeof(long long), hipMemcpyHostToDevice, st));
    return BAMM_OK;
}

static int shm_allreduce_i64(bamm_comm* c, void* dev_ptr, size_t n_words, hipStream_t st) {
    ShmGroup* g = c->shm;
    if (n_words > g->cap) { set_error("shared-memory all-reduce of %zu words (capacity %llu)", n_words, (unsigned long long)g->cap); return BAMM_ERR_ARG; }
    BAMM_HIP(hipSetDevice(ctx_device(c->ctx)));
    BAMM_HIP(hipMemcpyAsync(c->h_sum, dev_ptr, n_words * sizeof(long long), hipMemcpyDeviceToHost, st));
    BAMM_HIP(hipStreamSynchronize(st));
    memcpy(g->inbox(c->rank), c->h_sum, n_words * sizeof(long long));
    if (!g->barrier()) { set_error("shared-memory all-reduce: the group was aborted or a rank did not arrive"); return BAMM_ERR_COMM; }
    for (size_t i = 0; i < n_words; i++) {                   // integer sums: the rank order does not matter
        long long t = 0;
        for (uint32_t r = 0; r < g->n; r++) t += g->inbox(r)[i];
        c->h_sum[i] = t;
    }
    if (!g->barrier()) { set_error("shared-memory all-reduce: the group was aborted or a rank did not arrive"); return BAMM_ERR_COMM; }
    BAMM_HIP(hipMemcpyAsync(dev_ptr, c->h_sum, n_words * sizeof(long long), hipMemcpyHostToDevice, st));
    BAMM_HIP(hipStreamSynchronize(st));                      // h_sum is reused by the next call
    return BAMM_OK;
}

int comm_allreduce_i64(bamm_comm* c, void* dev_ptr, size_t n_words, hipStream_t st) {
    if (c->aborted.load(std::memory_order_acquire)) { set_error("the communicator was aborted"); return BAMM_ERR_COMM; }
    if (c->local) return local_allreduce_i64(c, dev_ptr, n_words, st);
    if (c->shm) return shm_allreduce_i64(c, dev_ptr, n_words, st);
    const Rccl* r = rccl();
    if (!r) return BAMM_ERR_COMM;
    std::lock_guard<std::mutex> lock(c->mu);
    if (!c->comm) { set_error("the communicator was aborted"); return BAMM_ERR_COMM; }
    const ncclResult_t rc = r->AllReduce(dev_ptr, dev_ptr, n_words, ncclInt64, ncclSum, c->comm, st);
    return rc == ncclSuccess ? BAMM_OK : fail(r, "ncclAllReduce", rc);
}

bamm_ctx* comm_ctx(const bamm_comm* c) { return c->ctx; }
bool comm_aborted(const bamm_comm* c) { return c->aborted.load(std::memory_order_acquire); }

// words summed over the communicator through a device buffer (set-up traffic: votes, pointers, IPC handles)
static int sum_words(bamm_comm* c, long long* h, size_t n) {
    hipStream_t st = ctx_stream(c->ctx);
    long long* d = nullptr;
    BAMM_HIP(hipSetDevice(ctx_device(c->ctx)));
    BAMM_HIP(hipMalloc((void**)&d, n * sizeof(long long)));
    hipError_t e = hipMemcpyAsync(d, h, n * sizeof(long long), hipMemcpyHostToDevice, st);
    int rc = BAMM_OK;
    if (e == hipSuccess) rc = comm_allreduce_i64(c, d, n, st);
    if (e == hipSuccess && !rc) e = hipMemcpyAsync(h, d, n * sizeof(long long), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess && !rc) e = hipStreamSynchronize(st);
    (void)hipFree(d);
    if (rc) return rc;
    if (e != hipSuccess) { set_error("communicator set-up traffic: %s", hipGetErrorString(e)); return BAMM_ERR_HIP; }
    return BAMM_OK;
}

// Who a rank of the table is: "same process" must mean the same address space, and a pid alone does not say that -- ranks in
// different pid namespaces or on different hosts (containers commonly all run as pid 1..20; an RCCL communicator may span
// nodes) can carry equal pids.  A rank is this process only if the host (hostname + boot id + pid namespace), the pid AND a
// nonce drawn once per process all match; everything else goes through hipIpcOpenMemHandle, which fails cleanly across hosts.
struct ProcessId { long long host, nonce; };
static const ProcessId& process_id() {
    static const ProcessId id = [] {
        auto fnv = [](unsigned long long h, const char* p, size_t n) { for (size_t i = 0; i < n; i++) { h ^= (unsigned char)p[i]; h *= 1099511628211ull; } return h; };
        unsigned long long h = 1469598103934665603ull;
        char buf[256];
        if (gethostname(buf, sizeof buf) == 0) { buf[sizeof buf - 1] = 0; h = fnv(h, buf, strlen(buf)); }
        if (FILE* f = fopen("/proc/sys/kernel/random/boot_id", "r")) { const size_t n = fread(buf, 1, sizeof buf, f); fclose(f); h = fnv(h, buf, n); }
        const ssize_t n = readlink("/proc/self/ns/pid", buf, sizeof buf);
        if (n > 0) h = fnv(h, buf, (size_t)n);
        unsigned long long r = 0;
        if (FILE* f = fopen("/dev/urandom", "rb")) { if (fread(&r, sizeof r, 1, f) != 1) r = 0; fclose(f); }
        r ^= (unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count() * 0x9e3779b97f4a7c15ull ^ ((unsigned long long)getpid() << 32);
        return ProcessId{(long long)(h >> 2), (long long)(r >> 2) | 1ll};          // non-negative: the table travels as a sum
    }();
    return id;
}

static void peer_release(bamm_comm* c) {
    for (uint32_t r = 0; r < kPeerMaxWorld; r++) {
        if (c->peer_map[r] && c->peer_ipc[r]) (void)hipIpcCloseMemHandle(c->peer_map[r]);
        c->peer_map[r] = nullptr; c->peer_ipc[r] = false;
    }
    if (c->peer_inbox) (void)hipFree(c->peer_inbox);
    c->peer_inbox = nullptr;
    c->peer_ready = false;
}

// Collective (every rank calls it, once per communicator; later calls return the first outcome): inboxes of
// 3 x world x stride_words entries of 16 bytes in fine-grained device memory, mapped into every peer -- the pointer itself for
// ranks of this process (peer access enabled between their devices), an IPC handle for ranks in other processes;
// both travel over the communicator's own all-reduce (each rank fills its row of a zeroed table: the sum is the
// all-gather).  The ranks then vote: the inboxes are used only if EVERY rank mapped all of them.  A refusal is not an
// error: *ready = 0, comm_peer_why() says why, and the caller stays on the RCCL collective.
int comm_peer_setup(bamm_comm* c, uint32_t stride_words, int* ready) {
    *ready = 0;
    if (c->peer_tried) {
        *ready = (c->peer_ready && stride_words <= c->peer_stride) ? 1 : 0;
        return BAMM_OK;
    }
    c->peer_tried = true;
    const uint32_t world = c->world, me = c->rank;
    if (world < 2u || world > kPeerMaxWorld) { c->peer_why = "the in-kernel all-reduce serves 2..8 ranks"; return BAMM_OK; }
    BAMM_HIP(hipSetDevice(ctx_device(c->ctx)));
    const size_t words = (size_t)3 * world * stride_words * 2;       // an entry = 16 bytes
    bool ok = true;
    std::string why;
    hipIpcMemHandle_t handle;
    memset(&handle, 0, sizeof handle);
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "IPC handle size");
    if (hipExtMallocWithFlags((void**)&c->peer_inbox, words * sizeof(long long), hipDeviceMallocFinegrained) != hipSuccess) {
        (void)hipGetLastError();
        ok = false; why = "fine-grained device memory could not be allocated";
        c->peer_inbox = nullptr;
    }
    if (ok && hipMemset(c->peer_inbox, 0, words * sizeof(long long)) != hipSuccess) { ok = false; why = "hipMemset of the inbox failed"; }
    bool have_handle = false;
    if (ok) {
        have_handle = hipIpcGetMemHandle(&handle, c->peer_inbox) == hipSuccess;
        if (!have_handle) (void)hipGetLastError();           // only needed by ranks in other processes (checked below)
    }
    // row r of the table: [pid, device, pointer, handle present, 8 words of IPC handle, stride, host id, process nonce]
    constexpr size_t ROW = 15;
    std::vector<long long> tab((size_t)world * ROW, 0);
    long long* mine = tab.data() + (size_t)me * ROW;
    mine[0] = (long long)getpid(); mine[1] = ctx_device(c->ctx); mine[2] = (long long)(uintptr_t)c->peer_inbox;
    mine[3] = have_handle ? 1 : 0;
    memcpy(mine + 4, &handle, sizeof handle);
    mine[12] = stride_words;
    mine[13] = process_id().host; mine[14] = process_id().nonce;
    int rc = sum_words(c, tab.data(), tab.size());
    if (rc) { peer_release(c); return rc; }
    for (uint32_t r = 0; r < world && ok; r++) {
        const long long* row = tab.data() + (size_t)r * ROW;
        if (row[12] != (long long)stride_words) { ok = false; why = "the ranks ask for inboxes of different sizes"; break; }
        if (r == me) continue;
        if (row[2] == 0) { ok = false; why = "a peer has no inbox"; break; }
        if (row[0] == (long long)getpid() && row[13] == process_id().host && row[14] == process_id().nonce) {   // a rank of THIS process: its pointer is valid here
            if ((int)row[1] != ctx_device(c->ctx)) {
                const hipError_t e = hipDeviceEnablePeerAccess((int)row[1], 0);
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) { ok = false; why = std::string("hipDeviceEnablePeerAccess: ") + hipGetErrorString(e); }
                (void)hipGetLastError();
            }
            c->peer_map[r] = (void*)(uintptr_t)row[2];
        } else {
            if (!row[3]) { ok = false; why = "a peer in another process could not export its inbox (hipIpcGetMemHandle)"; break; }
            hipIpcMemHandle_t h;
            memcpy(&h, row + 4, sizeof h);
            void* p = nullptr;
            const hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
            if (e != hipSuccess) { (void)hipGetLastError(); ok = false; why = std::string("hipIpcOpenMemHandle: ") + hipGetErrorString(e); break; }
            c->peer_map[r] = p;
            c->peer_ipc[r] = true;
        }
    }
    long long vote[1] = {ok ? 1 : 0};
    rc = sum_words(c, vote, 1);
    if (rc) { peer_release(c); return rc; }
    if (vote[0] != (long long)world) {
        c->peer_why = ok ? "a peer could not map the inboxes" : why;
        peer_release(c);
        return BAMM_OK;
    }
    c->peer_stride = stride_words;
    c->peer_ready = true;
    *ready = 1;
    return BAMM_OK;
}

const char* comm_peer_why(const bamm_comm* c) { return c->peer_why.c_str(); }

// the device-side view for one launch; the caller fills the slots / sequence numbers / ticket / err
void comm_peer_args(const bamm_comm* c, PeerArgs* p) {
    *p = PeerArgs{};
    p->world = c->world; p->rank = c->rank; p->stride = c->peer_stride; p->inbox = c->peer_inbox;
    for (uint32_t r = 0; r < c->world; r++) p->peer[r] = c->peer_map[r];
}

unsigned long long comm_peer_next_seq(bamm_comm* c) {
    if ((uint32_t)(++c->peer_seq) == 0u) ++c->peer_seq;       // the low 32 bits tag the entries: never the zero of a fresh inbox
    return c->peer_seq;
}

}  // namespace bamm

extern "C" {

int bamm_comm_init_all(bamm_ctx* const* ctxs, uint32_t n, bamm_comm** out) {
    if (!ctxs || !out || n == 0) { set_error("bamm_comm_init_all: bad argument"); return BAMM_ERR_ARG; }
    for (uint32_t i = 0; i < n; i++) out[i] = nullptr;
    const Rccl* r = rccl();
    if (!r) return BAMM_ERR_COMM;
    std::vector<int> devs(n);
    for (uint32_t i = 0; i < n; i++) {
        if (!ctxs[i]) { set_error("bamm_comm_init_all: null context %u", i); return BAMM_ERR_ARG; }
        devs[i] = ctx_device(ctxs[i]);
        for (uint32_t j = 0; j < i; j++)
            if (devs[j] == devs[i]) {
                set_error("bamm_comm_init_all: device %d appears twice (one rank per GPU)", devs[i]);
                return BAMM_ERR_ARG;
            }
    }
    std::vector<ncclComm_t> comms(n, nullptr);
    const ncclResult_t rc = r->CommInitAll(comms.data(), (int)n, devs.data());
    if (rc != ncclSuccess) return fail(r, "ncclCommInitAll", rc);
    for (uint32_t i = 0; i < n; i++) {
        bamm_comm* c = new bamm_comm();
        c->comm = comms[i]; c->ctx = ctxs[i]; c->rank = i; c->world = n;
        out[i] = c;
    }
    return BAMM_OK;
}

int bamm_comm_init_local(bamm_ctx* const* ctxs, uint32_t n, uint64_t max_words, bamm_comm** out) {
    if (!ctxs || !out || n == 0 || max_words == 0) { set_error("bamm_comm_init_local: bad argument"); return BAMM_ERR_ARG; }
    for (uint32_t i = 0; i < n; i++) out[i] = nullptr;
    for (uint32_t i = 0; i < n; i++)
        if (!ctxs[i]) { set_error("bamm_comm_init_local: null context %u", i); return BAMM_ERR_ARG; }
    LocalGroup* g = new LocalGroup();
    g->n = n; g->cap = (size_t)max_words; g->refs = n;
    g->inbox.assign(n, nullptr);
    std::vector<bamm_comm*> made;
    bool ok = true;
    for (uint32_t i = 0; i < n && ok; i++) {
        bamm_comm* c = new bamm_comm();
        c->ctx = ctxs[i]; c->rank = i; c->world = n; c->local = g;
        made.push_back(c);
        ok = hipSetDevice(ctx_device(ctxs[i])) == hipSuccess &&
             hipHostMalloc((void**)&g->inbox[i], g->cap * sizeof(long long), hipHostMallocPortable) == hipSuccess &&
             hipHostMalloc((void**)&c->h_sum, g->cap * sizeof(long long), hipHostMallocPortable) == hipSuccess;
    }
    if (!ok) {
        for (bamm_comm* c : made) { (void)hipHostFree(c->h_sum); delete c; }
        for (long long* p : g->inbox) (void)hipHostFree(p);
        delete g;
        set_error("bamm_comm_init_local: pinned host buffers could not be allocated");
        return BAMM_ERR_HIP;
    }
    for (uint32_t i = 0; i < n; i++) out[i] = made[i];
    return BAMM_OK;
}

int bamm_comm_init_shm(bamm_ctx* ctx, const char* name, uint32_t rank, uint32_t world, uint64_t max_words, bamm_comm** out) {
    if (!ctx || !name || !out || world == 0 || rank >= world || max_words == 0 || name[0] != '/' || world > kShmMaxWorld) {
        set_error("bamm_comm_init_shm: bad argument (the name starts with '/', at most %u ranks)", kShmMaxWorld);
        return BAMM_ERR_ARG;
    }
    *out = nullptr;
    const size_t bytes = sizeof(ShmGroup) + (size_t)world * (size_t)max_words * sizeof(long long);
    const auto t0 = std::chrono::steady_clock::now();
    auto elapsed = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
    // Rank 0 -- and nobody else -- creates the segment, after removing whatever carries the name (a creator killed before
    // its shm_unlink leaves the name behind with a stale barrier state in it; names are recycled with pids).  An attacher
    // cannot tell a stale segment from the new one by looking at it, so it does not try: it writes a fresh nonce into its
    // hello slot and only a LIVE rank 0 answers (ack = hello), which it does in the segment it created and in no other.  No
    // answer within a moment: unmap, open the name again (by then rank 0 has replaced it).
    ShmGroup* g = nullptr;
    if (rank == 0) {
        (void)shm_unlink(name);
        const int fd = shm_open(name, O_RDWR | O_CREAT | O_EXCL, 0600);
        if (fd < 0) { set_error("bamm_comm_init_shm: cannot create %s", name); return BAMM_ERR_COMM; }
        if (ftruncate(fd, (off_t)bytes) != 0) { close(fd); shm_unlink(name); set_error("bamm_comm_init_shm: cannot size %s", name); return BAMM_ERR_COMM; }
        void* m = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        close(fd);
        if (m == MAP_FAILED) { shm_unlink(name); set_error("bamm_comm_init_shm: mmap failed"); return BAMM_ERR_COMM; }
        g = reinterpret_cast<ShmGroup*>(m);
        g->arrived.store(0); g->generation.store(0); g->aborted.store(0); g->attached.store(0);
        for (uint32_t r = 0; r < kShmMaxWorld; r++) { g->hello[r].store(0); g->ack[r].store(0); }
        g->n = world; g->cap = max_words;
        g->magic.store(kShmMagic, std::memory_order_release);
        // answer the attachers until all of them have joined (each counts itself in `attached` once it has seen its answer; one
        // that opened a stale segment first comes back with a new nonce)
        while (g->attached.load(std::memory_order_acquire) + 1u < world) {
            for (uint32_t r = 1; r < world; r++) {
                const unsigned long long h = g->hello[r].load(std::memory_order_acquire);
                if (h != 0ull && g->ack[r].load(std::memory_order_relaxed) != h) g->ack[r].store(h, std::memory_order_release);
            }
            if (elapsed() > 30.0) { g->aborted.store(1, std::memory_order_release); munmap(m, bytes); shm_unlink(name); set_error("bamm_comm_init_shm: the other ranks did not attach to %s", name); return BAMM_ERR_COMM; }
            std::this_thread::sleep_for(std::chrono::microseconds(100));
        }
    } else {
        unsigned long long nonce = (unsigned long long)process_id().nonce ^ ((unsigned long long)rank << 56);
        std::string why = "rank 0 did not create it";
        while (!g && elapsed() < 30.0) {
            const int fd = shm_open(name, O_RDWR, 0600);
            if (fd < 0) { std::this_thread::sleep_for(std::chrono::milliseconds(2)); continue; }
            struct stat sb;
            if (fstat(fd, &sb) != 0 || (size_t)sb.st_size != bytes) {      // not sized yet, or another group's
                close(fd); why = "it has another size (another world / capacity?)";
                std::this_thread::sleep_for(std::chrono::milliseconds(2)); continue;
            }
            void* m = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
            close(fd);
            if (m == MAP_FAILED) { set_error("bamm_comm_init_shm: mmap failed"); return BAMM_ERR_COMM; }
            ShmGroup* cand = reinterpret_cast<ShmGroup*>(m);
            const auto t1 = std::chrono::steady_clock::now();
            auto waited = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count(); };
            while (cand->magic.load(std::memory_order_acquire) != kShmMagic && waited() < 0.2) std::this_thread::sleep_for(std::chrono::microseconds(200));
            bool ok = cand->magic.load(std::memory_order_acquire) == kShmMagic && cand->n == world && cand->cap == max_words;
            if (ok) {
                nonce = nonce * 6364136223846793005ull + 1442695040888963407ull;       // a new one per attempt, never 0
                if (nonce == 0ull) nonce = 1ull;
                cand->hello[rank].store(nonce, std::memory_order_release);
                while (cand->ack[rank].load(std::memory_order_acquire) != nonce && waited() < 0.5) std::this_thread::sleep_for(std::chrono::microseconds(200));
                ok = cand->ack[rank].load(std::memory_order_acquire) == nonce;
            }
            if (ok) { g = cand; g->attached.fetch_add(1, std::memory_order_acq_rel); break; }
            why = "nobody answered in it (a segment left behind by an earlier run?)";
            munmap(m, bytes);
            std::this_thread::sleep_for(std::chrono::milliseconds(5));
        }
        if (!g) { set_error("bamm_comm_init_shm: cannot join %s: %s", name, why.c_str()); return BAMM_ERR_COMM; }
    }
    void* m = g;
    bamm_comm* c = new bamm_comm();
    c->ctx = ctx; c->rank = rank; c->world = world; c->shm = g; c->shm_bytes = bytes; c->shm_name = name;
    if (hipSetDevice(ctx_device(ctx)) != hipSuccess ||
        hipHostMalloc((void**)&c->h_sum, (size_t)max_words * sizeof(long long), hipHostMallocDefault) != hipSuccess) {
        g->aborted.store(1, std::memory_order_release);
        munmap(m, bytes); delete c;
        if (rank == 0) shm_unlink(name);
        set_error("bamm_comm_init_shm: pinned host buffer could not be allocated");
        return BAMM_ERR_HIP;
    }
    if (rank == 0) g->attached.fetch_add(1, std::memory_order_acq_rel);
    if (!g->barrier(30.0)) {                                 // everybody is attached before the name goes away
        (void)hipHostFree(c->h_sum); munmap(m, bytes); delete c;
        if (rank == 0) shm_unlink(name);
        set_error("bamm_comm_init_shm: the other ranks did not attach to %s", name);
        return BAMM_ERR_COMM;
    }
    if (rank == 0) shm_unlink(name);                         // the mappings keep the segment alive; no name is left behind
    *out = c;
    return BAMM_OK;
}

int bamm_comm_abort(bamm_comm* c) {
    if (!c) return BAMM_OK;
    c->aborted.store(true, std::memory_order_release);       // whatever happens below: this rank's later calls fail
    if (c->shm) { c->shm->aborted.store(1, std::memory_order_release); return BAMM_OK; }
    if (c->local) {
        { std::lock_guard<std::mutex> lock(c->local->mu); c->local->aborted = true; }
        c->local->cv.notify_all();
        return BAMM_OK;
    }
    const Rccl* r = rccl();
    ncclComm_t mine = nullptr;
    {   // take the handle out under the lock: of any number of threads aborting this communicator one gets it
        std::lock_guard<std::mutex> lock(c->mu);
        mine = c->comm;
        c->comm = nullptr;
    }
    if (r && mine) {
        (void)hipSetDevice(ctx_device(c->ctx));
        (void)r->CommAbort(mine);                            // also frees the communicator
    }
    return BAMM_OK;
}

int bamm_comm_time_allreduce(bamm_comm* c, uint64_t n_words, uint32_t iters, float* us_per_call) {
    if (!c || !us_per_call || n_words == 0 || iters == 0) { set_error("bamm_comm_time_allreduce: bad argument"); return BAMM_ERR_ARG; }
    *us_per_call = 0.0f;
    BAMM_HIP(hipSetDevice(ctx_device(c->ctx)));
    hipStream_t st = ctx_stream(c->ctx);
    long long* d = nullptr;
    BAMM_HIP(hipMalloc((void**)&d, n_words * sizeof(long long)));
    hipEvent_t a = nullptr, b = nullptr;
    int rc = BAMM_OK;
    hipError_t e = hipMemsetAsync(d, 0, n_words * sizeof(long long), st);
    if (e == hipSuccess) e = hipEventCreate(&a);
    if (e == hipSuccess) e = hipEventCreate(&b);
    const uint32_t warm = std::min(iters, 20u);
    for (uint32_t i = 0; i < warm && e == hipSuccess && !rc; i++) rc = comm_allreduce_i64(c, d, (size_t)n_words, st);
    if (e == hipSuccess && !rc) e = hipEventRecord(a, st);
    for (uint32_t i = 0; i < iters && e == hipSuccess && !rc; i++) rc = comm_allreduce_i64(c, d, (size_t)n_words, st);
    if (e == hipSuccess && !rc) e = hipEventRecord(b, st);
    if (e == hipSuccess && !rc) e = hipEventSynchronize(b);
    float ms = 0.0f;
    if (e == hipSuccess && !rc) e = hipEventElapsedTime(&ms, a, b);
    if (a) (void)hipEventDestroy(a);
    if (b) (void)hipEventDestroy(b);
    (void)hipFree(d);
    if (rc) return rc;
    if (e != hipSuccess) { set_error("bamm_comm_time_allreduce: %s", hipGetErrorString(e)); return BAMM_ERR_HIP; }
    *us_per_call = ms * 1e3f / (float)iters;
    return BAMM_OK;
}

int bamm_comm_unique_id(void* id_out, size_t cap) {
    if (!id_out || cap < BAMM_COMM_ID_BYTES) { set_error("bamm_comm_unique_id: buffer of %d bytes needed", BAMM_COMM_ID_BYTES); return BAMM_ERR_ARG; }
    static_assert(sizeof(ncclUniqueId) == BAMM_COMM_ID_BYTES, "ncclUniqueId size");
    const Rccl* r = rccl();
    if (!r) return BAMM_ERR_COMM;
    ncclUniqueId id;
    const ncclResult_t rc = r->GetUniqueId(&id);
    if (rc != ncclSuccess) return fail(r, "ncclGetUniqueId", rc);
    memcpy(id_out, &id, sizeof id);
    return BAMM_OK;
}

int bamm_comm_init_rank(bamm_ctx* ctx, const void* id, uint32_t rank, uint32_t world, bamm_comm** out) {
    if (!ctx || !id || !out || world == 0 || rank >= world) { set_error("bamm_comm_init_rank: bad argument"); return BAMM_ERR_ARG; }
    *out = nullptr;
    const Rccl* r = rccl();
    if (!r) return BAMM_ERR_COMM;
    BAMM_HIP(hipSetDevice(ctx_device(ctx)));
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof uid);
    ncclComm_t comm = nullptr;
    const ncclResult_t rc = r->CommInitRank(&comm, (int)world, uid, (int)rank);
    if (rc != ncclSuccess) return fail(r, "ncclCommInitRank", rc);
    bamm_comm* c = new bamm_comm();
    c->comm = comm; c->ctx = ctx; c->rank = rank; c->world = world;
    *out = c;
    return BAMM_OK;
}

int bamm_comm_info(const bamm_comm* c, uint32_t* rank, uint32_t* world, int* rccl_version) {
    if (!c) { set_error("null communicator"); return BAMM_ERR_ARG; }
    if (rank) *rank = c->rank;
    if (world) *world = c->world;
    if (rccl_version) {
        *rccl_version = 0;                                   // 0: the host-staged groups of bamm_comm_init_local / _shm
        if (!c->local && !c->shm) {
            const Rccl* r = rccl();
            if (r) (void)r->GetVersion(rccl_version);
        }
    }
    return BAMM_OK;
}

int bamm_comm_destroy(bamm_comm* c) {
    if (!c) return BAMM_OK;
    if (c->shm) {
        (void)hipSetDevice(ctx_device(c->ctx));
        peer_release(c);
        (void)hipHostFree(c->h_sum);
        munmap(c->shm, c->shm_bytes);
        delete c;
        return BAMM_OK;
    }
    if (c->local) {
        LocalGroup* g = c->local;
        (void)hipSetDevice(ctx_device(c->ctx));
        peer_release(c);
        (void)hipHostFree(c->h_sum);
        bool last;
        { std::lock_guard<std::mutex> lock(g->mu); last = --g->refs == 0; }
        if (last) {
            for (long long* p : g->inbox) (void)hipHostFree(p);
            delete g;
        }
        delete c;
        return BAMM_OK;
    }
    const Rccl* r = rccl();
    ncclComm_t mine = nullptr;
    {
        std::lock_guard<std::mutex> lock(c->mu);
        mine = c->comm;
        c->comm = nullptr;
    }
    (void)hipSetDevice(ctx_device(c->ctx));
    peer_release(c);
    if (r && mine) (void)r->CommDestroy(mine);
    delete c;
    return BAMM_OK;
}

}  // extern "C"
