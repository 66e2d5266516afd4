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
#include <rccl/rccl.h>

#include <atomic>
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

struct bamm_comm {
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

int comm_allreduce_i64(bamm_comm* c, void* dev_ptr, size_t n_words, hipStream_t st) {
    if (c->aborted.load(std::memory_order_acquire)) { set_error("the communicator was aborted"); return BAMM_ERR_COMM; }
    if (c->local) return local_allreduce_i64(c, dev_ptr, n_words, st);
    const Rccl* r = rccl();
    if (!r) return BAMM_ERR_COMM;
    std::lock_guard<std::mutex> lock(c->mu);
    if (!c->comm) { set_error("the communicator was aborted"); return BAMM_ERR_COMM; }
    const ncclResult_t rc = r->AllReduce(dev_ptr, dev_ptr, n_words, ncclInt64, ncclSum, c->comm, st);
    return rc == ncclSuccess ? BAMM_OK : fail(r, "ncclAllReduce", rc);
}

bamm_ctx* comm_ctx(const bamm_comm* c) { return c->ctx; }
bool comm_aborted(const bamm_comm* c) { return c->aborted.load(std::memory_order_acquire); }

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

int bamm_comm_abort(bamm_comm* c) {
    if (!c) return BAMM_OK;
    c->aborted.store(true, std::memory_order_release);       // whatever happens below: this rank's later calls fail
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
        *rccl_version = 0;                                   // 0: the host-staged group of bamm_comm_init_local
        if (!c->local) {
            const Rccl* r = rccl();
            if (r) (void)r->GetVersion(rccl_version);
        }
    }
    return BAMM_OK;
}

int bamm_comm_destroy(bamm_comm* c) {
    if (!c) return BAMM_OK;
    if (c->local) {
        LocalGroup* g = c->local;
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
    if (r && mine) {
        (void)hipSetDevice(ctx_device(c->ctx));
        (void)r->CommDestroy(mine);
    }
    delete c;
    return BAMM_OK;
}

}  // extern "C"
