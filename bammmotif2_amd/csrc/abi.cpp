// Device-side half of the C ABI (include/bamm_em.h): contexts, resident sequence sets, EM
// handles and the scorer.  Host code only -- kernels live in kernels.hip.
//
// Reference seam this replaces: class EM (/root/reference/src/refinement/EM.h:11-69,
// EM.cpp:7-259,505-527) and ScoreSeqSet::calcLogOdds (seq_scoring/ScoreSeqSet.cpp:25-67).

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <unordered_map>

#include <thread>

#include "common.h"
#include "glibc_rand.h"
#include "negs.h"
#include "prep.h"

// the longest length class (positions per lane) whose grouped kernels are built with the fused-update prologue and the
// all-reduce tail (grouped_kernel.h carries the same default)
#ifndef BAMM_FUSE_MAX_M
#define BAMM_FUSE_MAX_M 16
#endif

using namespace bamm;

static bool flush_idle_scratch(int device);
static void par_memcpy(void* dst, const void* src, size_t bytes);

namespace {

// hipMalloc; when the device is out of memory the contexts' idle scratch blocks (below) are released and it is tried again
int dev_alloc_bytes(void** p, size_t bytes) {
    *p = nullptr;
    hipError_t e = hipMalloc(p, bytes ? bytes : 1);
    if (e == hipErrorOutOfMemory) {
        (void)hipGetLastError();
        int device = 0;
        (void)hipGetDevice(&device);
        if (flush_idle_scratch(device)) e = hipMalloc(p, bytes ? bytes : 1);
    }
    if (e != hipSuccess) {
        set_error("hipMalloc of %zu bytes failed: %s", bytes, hipGetErrorString(e));
        *p = nullptr;
        return BAMM_ERR_HIP;
    }
    return BAMM_OK;
}

template <class T>
int dev_alloc(T** p, size_t count) { return dev_alloc_bytes((void**)p, count * sizeof(T)); }

// a std::vector whose resize() leaves the new elements uninitialised: the set-sized host mirrors are filled by a parallel
// copy right after (a value-initialising resize is one more single-threaded pass over 100 MB)
template <class T>
struct NoInitAlloc : std::allocator<T> {
    template <class U> struct rebind { using other = NoInitAlloc<U>; };
    template <class U, class... A> void construct(U* p, A&&... a) {
        if constexpr (sizeof...(A) == 0) ::new ((void*)p) U; else ::new ((void*)p) U(std::forward<A>(a)...);
    }
};
template <class T> using RawVec = std::vector<T, NoInitAlloc<T>>;

struct Bucket {
    int mclass = 0;
    uint32_t count = 0;
    uint32_t* d_idx = nullptr;   // nullptr: all sequences in natural order
    std::vector<uint32_t> h_idx; // host copy of d_idx (empty with d_idx == nullptr)
    double work = 0;             // sum of M over the bucket (LDS instruction proxy)
};

struct ExcK {                    // exceptions relevant at one model order
    uint64_t* d_off = nullptr;
    uint2* d_exc = nullptr;
    uint64_t count = 0;
    RawVec<uint64_t> h_off;               // host copies (the grouped kernel's records are built from them)
    RawVec<uint2> h_ex;
    struct XRec {                         // grouped kernel (grouped.hip), one set per group size G
        uint4* d_xrec = nullptr;          // per-sequence record
        RawVec<uint8_t> h_B;              // group ends that need a virtual row (0 = no exception, 255 = too many)
        RawVec<uint32_t> h_lo;            // first of them
    };
    std::map<uint32_t, XRec> xrec;
};

struct EmBucket {                // one kernel launch of an EM pass
    int mclass = 0;
    uint32_t count = 0;
    const uint32_t* d_idx = nullptr;
    bool grouped = false;        // k_em_grp instead of k_em_seq
    uint32_t G = 0;              // its group size
    uint32_t layout = 0;         // table layout (grp_geometry)
    const uint4* d_xrec = nullptr;
    uint32_t blocks = 0, logc = 0, sparse_cap = 0, sparse_bytes = 0;
    double work = 0;
};

}  // namespace

struct bamm_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    uint32_t blocks = 0, threads = 0;   // 0 = default
    // bamm_ctx_set_tuning: kernel-selection switches for benchmarks and the cross-kernel parity tests
    bool use_grouped = true, use_sparse = true, use_e_fused = true, use_e_list = true, use_fused_update = true, use_adaptive_lists = true, use_update_blocks = true;
    uint32_t list_threshold_pct = 45;   // sliced path: a pass takes lists when fewer than this share of the windows was non-zero in the pass before
    uint32_t group_size = 0;            // 0 = planner's choice
    int group_layout = -1;              // -1 = planner's choice
    bool use_peer_allreduce = false;    // the pass's all-reduce inside the sequence kernels over peer-mapped inboxes (default off)
    uint32_t peer_timeout_ms = 2000;    // how long a block waits for a peer's sums before it gives up (BAMM_ERR_COMM)
    int num_cus = 0;
    std::string name;
    // Scratch that is as large as the sequence set (dense r, the lists between the E pass and the M slices, the fix
    // lanes' log, getR's staging): 10 GB per handle at config 4, and 0.1-0.3 s per handle to allocate and free on
    // some boxes of the pool (hipFree synchronises the device as well).  A handle returns such blocks to its context
    // and the next handle on it -- the next motif, the next CV fold -- takes them over; everything on a context runs
    // on its one stream, so the old owner's last kernel is ordered before the new owner's first.  At most a quarter
    // of the device's memory stays idle here, and an allocation that fails releases every context's idle blocks first.
    std::mutex scratch_mu;
    std::unordered_map<void*, size_t> scratch_live;          // blocks handed out: bytes
    std::vector<std::pair<void*, size_t>> scratch_idle;      // blocks waiting for their next owner, oldest first
    size_t scratch_idle_bytes = 0, scratch_cap_bytes = 0;
    bool scratch_poison = false;                             // tests: every block is filled with 0xFF when it is handed out
    uint64_t scratch_hits = 0, scratch_misses = 0;
    // Every transfer of 64 KiB or more between the CALLER's memory and the device goes through this pinned area (two chunks,
    // filled and drained in turn), never through a hipMemcpy on the caller's pages.  The HIP runtime pins pageable memory in
    // place for such a copy and keeps the registration; when the owner later unmaps those pages (a numpy array freed, a
    // std::vector going out of scope) the driver evicts the process's queues until it has dropped the registration:
    // the next launch or copy of the process then waits 17-39 ms (profiles/r05_first_call.txt -- the "29 ms second pass" of
    // profiles/r04_pass_times.txt was getR()'s result being freed; bamm_em_create's first upload paid the same for the
    // vectors of bamm_seqs_upload).  Memory this library pinned itself is never unmapped under a registration.
    std::mutex stage_mu;
    unsigned char* stage_buf[2] = {nullptr, nullptr};
    hipEvent_t stage_ev[2] = {nullptr, nullptr};
    bool stage_used[2] = {false, false};                     // an enqueued copy still reads (H2D) the chunk: wait for stage_ev first
};

constexpr size_t kScratchMinBytes = size_t(4) << 20;         // smaller blocks are plain allocations
static std::mutex g_ctx_mu;
static std::vector<bamm_ctx*> g_ctxs;                                // live contexts (flush_idle_scratch walks them)

static bool flush_idle_scratch(int device) {
    bool any = false;
    std::lock_guard<std::mutex> g(g_ctx_mu);
    for (bamm_ctx* c : g_ctxs) {
        if (c->device != device) continue;
        std::lock_guard<std::mutex> l(c->scratch_mu);
        for (auto& b : c->scratch_idle) { (void)hipFree(b.first); any = true; }
        c->scratch_idle.clear();
        c->scratch_idle_bytes = 0;
    }
    return any;
}

constexpr size_t kStageChunk = size_t(4) << 20;               // bytes per pinned chunk (two of them: 8 MB pinned per context, 1.5 ms to allocate)
constexpr size_t kStageMin = size_t(64) << 10;                // smaller transfers: the runtime copies them through its own staging buffer

// memcpy on the host threads the process was granted (a chunk of 8 MB: 0.9 ms on one thread, 0.2 ms on eight)
static void par_memcpy(void* dst, const void* src, size_t bytes) {
    const uint32_t T = (uint32_t)std::max<size_t>(1, std::min<size_t>(host_threads_hint(), bytes >> 20));
    if (T <= 1) { memcpy(dst, src, bytes); return; }
    std::vector<std::thread> th;
    for (uint32_t t = 0; t < T; t++) {
        const size_t b = bytes * t / T & ~size_t(63), e = t + 1 == T ? bytes : (bytes * (t + 1) / T & ~size_t(63));
        th.emplace_back([=] { memcpy((unsigned char*)dst + b, (const unsigned char*)src + b, e - b); });
    }
    for (auto& x : th) x.join();
}

static int stage_ready(bamm_ctx* c) {                         // stage_mu held
    if (c->stage_buf[0]) return BAMM_OK;
    unsigned char* p = nullptr;
    if (hipHostMalloc((void**)&p, 2 * kStageChunk, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        set_error("hipHostMalloc of the %zu-byte staging area failed", 2 * kStageChunk);
        return BAMM_ERR_HIP;
    }
    for (hipEvent_t& e : c->stage_ev)
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { (void)hipHostFree(p); set_error("hipEventCreate failed"); return BAMM_ERR_HIP; }
    c->stage_buf[0] = p; c->stage_buf[1] = p + kStageChunk;
    return BAMM_OK;
}

// host -> device on the context's stream.  Returns when `src` has been read (the caller may free it); the copies themselves
// are ordered on the stream like any other work.
int ctx_upload(bamm_ctx* c, void* dst_dev, const void* src, size_t bytes) {
    if (!bytes) return BAMM_OK;
    if (bytes < kStageMin) { BAMM_HIP(hipMemcpyAsync(dst_dev, src, bytes, hipMemcpyHostToDevice, c->stream)); return BAMM_OK; }
    std::lock_guard<std::mutex> l(c->stage_mu);
    if (int rc = stage_ready(c)) return rc;
    uint32_t b = 0;
    for (size_t at = 0; at < bytes; at += kStageChunk, b ^= 1u) {
        const size_t len = std::min(kStageChunk, bytes - at);
        if (c->stage_used[b]) BAMM_HIP(hipEventSynchronize(c->stage_ev[b]));      // the copy that last read this chunk is done
        par_memcpy(c->stage_buf[b], (const unsigned char*)src + at, len);
        BAMM_HIP(hipMemcpyAsync((unsigned char*)dst_dev + at, c->stage_buf[b], len, hipMemcpyHostToDevice, c->stream));
        BAMM_HIP(hipEventRecord(c->stage_ev[b], c->stream));
        c->stage_used[b] = true;
    }
    return BAMM_OK;
}

// device -> host on the context's stream.  Returns when `dst` holds the data (everything enqueued on the stream before it
// has completed by then).  Small transfers are enqueued only, like a hipMemcpyAsync: the caller synchronises.
int ctx_download(bamm_ctx* c, void* dst, const void* src_dev, size_t bytes) {
    if (!bytes) return BAMM_OK;
    if (bytes < kStageMin) { BAMM_HIP(hipMemcpyAsync(dst, src_dev, bytes, hipMemcpyDeviceToHost, c->stream)); return BAMM_OK; }
    std::lock_guard<std::mutex> l(c->stage_mu);
    if (int rc = stage_ready(c)) return rc;
    for (uint32_t b = 0; b < 2u; b++)
        if (c->stage_used[b]) { BAMM_HIP(hipEventSynchronize(c->stage_ev[b])); c->stage_used[b] = false; }
    const size_t chunks = (bytes + kStageChunk - 1) / kStageChunk;
    for (size_t i = 0; i <= chunks; i++) {                   // chunk i is enqueued while chunk i - 1 is copied out
        if (i < chunks) {
            const size_t at = i * kStageChunk, len = std::min(kStageChunk, bytes - at);
            BAMM_HIP(hipMemcpyAsync(c->stage_buf[i & 1u], (const unsigned char*)src_dev + at, len, hipMemcpyDeviceToHost, c->stream));
            BAMM_HIP(hipEventRecord(c->stage_ev[i & 1u], c->stream));
        }
        if (i > 0) {
            const size_t at = (i - 1) * kStageChunk, len = std::min(kStageChunk, bytes - at);
            BAMM_HIP(hipEventSynchronize(c->stage_ev[(i - 1) & 1u]));
            par_memcpy((unsigned char*)dst + at, c->stage_buf[(i - 1) & 1u], len);
        }
    }
    return BAMM_OK;
}

template <class T>
int dev_upload(bamm_ctx* c, T** p, const T* host, size_t count) {
    int rc = dev_alloc(p, count);
    if (rc) return rc;
    return ctx_upload(c, *p, host, count * sizeof(T));
}

template <class T>
int scratch_alloc(bamm_ctx* c, T** p, size_t count) {
    *p = nullptr;
    size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
    if (bytes < kScratchMinBytes) return dev_alloc(p, count);
    bytes = (bytes + (size_t(2) << 20) - 1) & ~((size_t(2) << 20) - 1);
    void* got = nullptr;
    size_t got_bytes = 0;
    {
        std::lock_guard<std::mutex> l(c->scratch_mu);
        size_t best = (size_t)-1;
        for (size_t i = 0; i < c->scratch_idle.size(); i++) {   // the tightest idle block of at most twice the size
            const size_t b = c->scratch_idle[i].second;
            if (b >= bytes && b <= 2 * bytes && (best == (size_t)-1 || b < c->scratch_idle[best].second)) best = i;
        }
        if (best != (size_t)-1) {
            got = c->scratch_idle[best].first; got_bytes = c->scratch_idle[best].second;
            c->scratch_idle.erase(c->scratch_idle.begin() + (ptrdiff_t)best);
            c->scratch_idle_bytes -= got_bytes;
            c->scratch_hits++;
        } else {
            c->scratch_misses++;
        }
    }
    if (!got) {
        if (int rc = dev_alloc_bytes(&got, bytes)) return rc;
        got_bytes = bytes;
    }
    if (c->scratch_poison && hipMemsetAsync(got, 0xff, got_bytes, c->stream) != hipSuccess) {
        (void)hipFree(got);
        set_error("hipMemsetAsync failed");
        return BAMM_ERR_HIP;
    }
    {
        std::lock_guard<std::mutex> l(c->scratch_mu);
        c->scratch_live[got] = got_bytes;
    }
    *p = (T*)got;
    return BAMM_OK;
}

// Returns a block to its context (the caller has made sure that nothing enqueued on OTHER streams still uses it;
// work on the context's own stream is ordered before the next owner's).  Plain allocations are freed.
void scratch_free(bamm_ctx* c, void* p) {
    if (!p) return;
    std::vector<void*> evict;
    {
        std::lock_guard<std::mutex> l(c->scratch_mu);
        auto it = c->scratch_live.find(p);
        if (it == c->scratch_live.end()) {
            evict.push_back(p);
        } else {
            c->scratch_idle.emplace_back(p, it->second);
            c->scratch_idle_bytes += it->second;
            c->scratch_live.erase(it);
            while (c->scratch_idle_bytes > c->scratch_cap_bytes && !c->scratch_idle.empty()) {     // oldest first
                evict.push_back(c->scratch_idle.front().first);
                c->scratch_idle_bytes -= c->scratch_idle.front().second;
                c->scratch_idle.erase(c->scratch_idle.begin());
            }
        }
    }
    for (void* q : evict) (void)hipFree(q);
}

struct bamm_seqs {
    bamm_ctx* ctx = nullptr;
    int refs = 1;
    std::mutex mu;                              // guards refs and the lazily built per-order tables (handles may be
                                                // created on one set from several host threads, FDR.cpp:37)
    uint64_t n = 0, total_len = 0;
    uint32_t max_len = 0, min_len = 0;
    uint64_t hbm_bytes = 0;
    uint32_t* d_words = nullptr;
    uint64_t* d_word_off = nullptr;
    uint32_t* d_len = nullptr;
    uint64_t* d_pos_off = nullptr;
    std::vector<uint32_t> h_len;
    RawVec<uint32_t> h_words;                   // host copy of the 2-bit stream (grouped kernel's exception records)
    std::vector<uint64_t> h_word_off;
    std::vector<uint64_t> h_pos_off;
    std::vector<uint64_t> h_exc_off;            // full (11-mer level) exception list
    RawVec<uint32_t> h_exc_pos, h_exc_kmer, h_exc_clean;
    std::vector<Bucket> buckets;
    std::map<uint32_t, ExcK> exc_by_order;      // node-based: pointers into it stay valid

    ~bamm_seqs() {                               // also runs on every error path of bamm_seqs_upload
        (void)hipFree(d_words);
        (void)hipFree(d_word_off);
        (void)hipFree(d_len);
        (void)hipFree(d_pos_off);
        for (auto& b : buckets) (void)hipFree(b.d_idx);
        for (auto& kv : exc_by_order) {
            (void)hipFree(kv.second.d_off); (void)hipFree(kv.second.d_exc);
            for (auto& x : kv.second.xrec) (void)hipFree(x.second.d_xrec);
        }
    }
};

struct EmBook {                   // host-side state one model update moves (optimize() rolls back work that did not happen)
    float *d_s, *d_s_alt, *d_q, *d_v, *d_v_alt; const float *s_last, *q_last;
    long long* d_acc; uint32_t acc_cur, llh_cur, host_iteration, events_used, pass_no;
    bool estep_done, acc_dirty, mask_done, ring_prev_dirty;
};

struct bamm_em {
    bamm_ctx* ctx = nullptr;
    bamm_seqs* seqs = nullptr;
    bamm_em_params prm{};
    uint32_t Y = 0, Kbg = 0;
    size_t vsz = 0, cells = 0;
    float *d_vbg = nullptr, *d_A = nullptr, *d_v = nullptr, *d_n = nullptr, *d_s = nullptr;
    float *d_q = nullptr, *d_status = nullptr, *d_trace = nullptr;
    // the odds table / q the most recent E pass used stay intact for getR() and MStep(): s is double
    // buffered (only the update writes it), q lives in three slots because EM::optimize_q() may write
    // it between any two of EStep / MStep / getR (EM.cpp:93-99,505-519) -- see q_write_slot()
    float *d_s_alt = nullptr;
    float *d_qbuf[3] = {nullptr, nullptr, nullptr};
    const float *s_last = nullptr, *q_last = nullptr;
    uint32_t* d_iteration = nullptr;
    uint8_t* d_mask = nullptr;
    // the pass's fused accumulator [cells | llh | sum_r | n_seqs]: 64-bit integers the blocks add into,
    // summed across ranks as int64 (exact, order-free), consumed and zeroed by the update
    long long* d_acc = nullptr;                // the slot the current / next pass adds into
    // ... a ring of three slots when the handle can fuse the model update into the next pass's kernel
    // (update_kernel.h): pass p adds into slot p mod 3, the next kernel's blocks read it, its writer block clears
    // the slot after next.  Outside a fused sequence only slot `acc_cur` is ever non-zero.
    long long* d_acc_ring = nullptr;
    size_t acc_stride = 0;                      // words per slot
    uint32_t acc_cur = 0;
    bool fusable = false;                       // K <= 2-sized tables, first launch of a pass is a grouped kernel with room for the update
    uint32_t fuse_upd_off = 0;                  // LDS offset of the update's scratch in that kernel
    float* d_s_block = nullptr;                 // [blocks of the first launch][W * (Y + 1)]
    float* d_v_alt = nullptr;                   // fused updates read the old v while the writer block stores the new one
    float* d_llh[2] = {nullptr, nullptr};       // log-likelihood of the last two updates (the stop rule compares them)
    double* d_upd_partial = nullptr;            // the update spread over blocks (tables beyond its LDS form): v_diff partials
    uint32_t* d_upd_ticket = nullptr;           // ... and the word its blocks draw tickets from
    uint32_t llh_cur = 0;                       // slot the last update wrote
    bool ring_prev_dirty = false;               // the ring slot behind acc_cur was read by a fused update and awaits clearing
    bool acc_external = false;                 // caller-owned (bamm_em_set_reduce_buffer)
    bool acc_dirty = false;                    // holds sums nobody consumed (accumulate without update, getR replay)
    uint32_t fix_shift = 40;                   // counts travel in units of 2^-fix_shift (40 unless the set is huge)
    float* h_status = nullptr;                  // pinned, 8 floats (+ 2 x 8 for optimize()'s look-ahead where the mirror below is missing)
    unsigned long long* h_tagged = nullptr;     // ... + 2 x 8 tagged words behind them: what the updates of optimize() report (UpdateArgs::status_mirror)
    unsigned long long* d_status_mirror = nullptr;   // h_tagged as the device addresses it
    uint32_t* d_stop = nullptr;                 // optimize(): set by k_update when the stop rule fires
    const uint32_t* stop_arg = nullptr;         // what the kernels are handed: d_stop inside optimize(), else null
    hipEvent_t opt_events[2] = {nullptr, nullptr};
    // this optimize() call's stop rule for k_update (run_update fills UpdateArgs from it)
    uint32_t opt_iteration = 0;
    float opt_llh_prev = 0.0f;
    uint32_t total_blocks = 0;
    std::vector<EmBucket> ebuckets;             // launches of one pass (length class x kernel flavour)
    std::vector<uint32_t*> owned_idx;           // index lists made for this handle (capable / other split)
    uint32_t threads = 0;
    // column-sliced path (tables beyond the fused kernel's LDS budget)
    bool sliced = false;
    std::vector<std::pair<uint32_t, uint32_t>> e_slices, m_slices;
    uint32_t m_slice_logc = 0;
    bool e_fused = false;                       // the E pass of the sliced path is k_em_seq (whole odds table in LDS)
    uint32_t m_slice_cap = 0;                   // sparse list capacity per wave in the M-slices (0 = dense)
    float* d_state = nullptr;                   // one float per position slot: E-chain state, then r (allocated on first use)
    // e_fused: the E pass hands the M slices compacted lists of the non-zero windows instead of dense r
    float* d_list_r = nullptr;
    uint16_t* d_list_p = nullptr;
    uint32_t* d_list_n = nullptr;
    // ... or dense r, chosen per pass on the device: [2] counts of windows with a non-zero addend (the pass before, this pass)
    unsigned long long* d_nnz = nullptr;
    uint32_t nnz_prev_slot = 0;
    unsigned long long nnz_limit = 0;           // above it a pass takes the dense flavour
    bool adaptive_lists = true;                 // bamm_ctx_set_tuning("adaptive_lists") when the handle was created
    // K = 3 through the grouped kernel: per-wave log of the virtual rows' counts (grouped_kernel.h), grown on demand
    unsigned long long* d_fix_log = nullptr;
    size_t fix_log_words = 0;
    ExcK* exc = nullptr;
    bool estep_done = false;
    float llh_prev = 0.0f;                      // EM.h:61
    uint32_t host_iteration = 0;
    bamm_allreduce_fn allreduce = nullptr;
    void* allreduce_user = nullptr;
    bamm_comm* comm = nullptr;                  // native RCCL all-reduce (bamm_em_set_comm)
    bool comm_verified = false;                 // verify_comm() ran with the peers
    // in-kernel all-reduce (PeerArgs): the last block of every accumulating pass exchanges the GPU's totals with the peers
    // and leaves the sum in the accumulator, in place -- no collective launch behind the pass
    bool peer_on = false;                       // agreed with every rank in verify_comm()
    bool pass_summed_in_kernel = false;         // the pass just enqueued carried the tail (launch_fused): run_allreduce has nothing to add
    uint32_t* d_peer_words = nullptr;           // [0] ticket, [1] err
    long long* d_comm_words = nullptr;          // four words for verify_comm()'s own sums, kept for the handle's life: a hipFree
                                                // there would synchronise the DEVICE, and with several ranks on one device (the
                                                // rehearsal forms) a peer that has already launched its first pass spins in that
                                                // kernel's tail for THIS rank's sums, which this rank cannot launch from inside hipFree
    std::string peer_note;                      // why peer_on is false although asked for
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
    uint32_t events_used = 0;
    uint32_t timing_every = 8, pass_no = 0;     // bamm_em_set_kernel_timing
    bool timing_now = false;
    // timing_every == BAMM_TIMING_WHOLE_CALL: ONE pair of events around all the passes of a call
    bool region_open = false;
    uint32_t region_passes = 0;
    std::vector<uint32_t> event_passes;          // passes between the two events of pair i (1 in the per-pass modes)
    // EM::mask state (allocated on first use)
    uint64_t n_active = 0;                      // sequences the handle trains on (mask applied)
    float* d_mask_r = nullptr;                  // responsibilities in the reference layout
    uint32_t* d_mask_bits = nullptr;
    long long* d_mask_hist = nullptr;
    MaskSelect* d_mask_sel = nullptr;
    float* d_mask_qseq = nullptr;
    unsigned long long* d_mask_partial_n = nullptr;
    double* d_mask_partial_stat = nullptr;
    uint32_t mask_blocks = 0;
    bool mask_done = false;                     // getR() serves d_mask_r
    EmBook books[4] = {};                       // snapshot right after update i at [i & 3]
};

namespace {

SeqView make_view(const bamm_seqs* s, const ExcK* exc, const Bucket& b, const uint8_t* d_mask) {
    SeqView v;
    v.words = s->d_words;
    v.word_off = s->d_word_off;
    v.len = s->d_len;
    v.pos_off = s->d_pos_off;
    v.exc_off = exc->d_off;
    v.exc = exc->d_exc;
    v.mask = d_mask;
    v.idx = b.d_idx;
    v.count = b.count;
    return v;
}

SeqView make_view(const bamm_seqs* s, const ExcK* exc, const EmBucket& b, const uint8_t* d_mask) {
    Bucket t;
    t.d_idx = const_cast<uint32_t*>(b.d_idx);
    t.count = b.count;
    return make_view(s, exc, t, d_mask);
}

// fn(begin, end) over contiguous ranges of [0, n) on the host threads the process was granted (bamm_set_host_threads):
// the per-sequence set-up loops below walk a million records and several million exceptions
template <class F>
void host_ranges(uint64_t n, F&& fn) {
    const uint32_t T = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(host_threads_hint(), n / 16384 + 1));
    if (T <= 1) { fn(uint64_t(0), n); return; }
    std::vector<std::thread> th;
    for (uint32_t t = 0; t < T; t++) th.emplace_back([&fn, n, t, T] { fn(n * t / T, n * (t + 1) / T); });
    for (auto& x : th) x.join();
}

// build (once per order) the list of positions whose kmer_ mod 4^(K+1) differs from what the
// 2-bit stream gives
int exceptions_for_order(bamm_seqs* s, uint32_t K, ExcK** out) {
    std::lock_guard<std::mutex> lock(s->mu);
    auto it = s->exc_by_order.find(K);
    if (it != s->exc_by_order.end()) { *out = &it->second; return BAMM_OK; }
    const uint32_t maskY = (uint32_t)(ipow4(K + 1) - 1);
    ExcK k;
    k.h_off.resize(s->n + 1);
    // two passes over fixed parts of the set, a thread each: counts per sequence (left in h_off[n + 1]) and per part, a scan over
    // the parts, then every part turns its counts into offsets while it fills its stretch of the list -- the same list in the
    // same order as one walk would give, with no pass over a million records on one thread
    const uint32_t parts = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(host_threads_hint(), s->n / 16384 + 1));
    std::vector<uint64_t> part_total(parts + 1, 0);
    auto part_range = [&](uint32_t t, uint64_t& n0, uint64_t& n1) { n0 = s->n * t / parts; n1 = s->n * (t + 1) / parts; };
    auto on_parts = [&](auto&& fn) {
        if (parts == 1) { fn(0u); return; }
        std::vector<std::thread> th;
        for (uint32_t t = 0; t < parts; t++) th.emplace_back([&fn, t] { fn(t); });
        for (auto& x : th) x.join();
    };
    on_parts([&](uint32_t t) {
        uint64_t n0, n1, total = 0;
        part_range(t, n0, n1);
        for (uint64_t n = n0; n < n1; n++) {
            uint64_t c = 0;
            for (uint64_t e = s->h_exc_off[n]; e < s->h_exc_off[n + 1]; e++) c += ((s->h_exc_kmer[e] ^ s->h_exc_clean[e]) & maskY) != 0u;
            k.h_off[n + 1] = c;
            total += c;
        }
        part_total[t + 1] = total;
    });
    for (uint32_t t = 0; t < parts; t++) part_total[t + 1] += part_total[t];
    k.h_ex.resize(part_total[parts]);
    k.h_off[0] = 0;
    on_parts([&](uint32_t t) {
        uint64_t n0, n1;
        part_range(t, n0, n1);
        uint64_t at = part_total[t];
        for (uint64_t n = n0; n < n1; n++) {
            for (uint64_t e = s->h_exc_off[n]; e < s->h_exc_off[n + 1]; e++)
                if (((s->h_exc_kmer[e] ^ s->h_exc_clean[e]) & maskY) != 0u)
                    k.h_ex[at++] = make_uint2(s->h_exc_pos[e], s->h_exc_kmer[e] & maskY);
            k.h_off[n + 1] = at;                              // (was the count: read above, by this thread)
        }
    });
    k.count = k.h_ex.size();
    int rc = dev_upload(s->ctx, &k.d_off, k.h_off.data(), k.h_off.size());
    if (rc) return rc;
    rc = dev_upload(s->ctx, &k.d_exc, k.h_ex.data(), k.h_ex.size());
    if (rc) { (void)hipFree(k.d_off); return rc; }
    if (hipStreamSynchronize(s->ctx->stream) != hipSuccess) {
        (void)hipFree(k.d_off); (void)hipFree(k.d_exc);
        set_error("stream sync failed while uploading the exception list");
        return BAMM_ERR_HIP;
    }
    auto ins = s->exc_by_order.emplace(K, std::move(k));
    *out = &ins.first->second;
    return BAMM_OK;
}

// records of the grouped kernel for group size G (built once per (order, G)): x = first exception
// position | B << 12, y/z/w = exact y of the positions lo-G+1 .. lo+B-1, one bit string of 7-bit fields (10 at K = 3)
int xrec_for_group(bamm_seqs* s, uint32_t K, uint32_t G, ExcK* k, const ExcK::XRec** out) {
    std::lock_guard<std::mutex> lock(s->mu);
    auto it = k->xrec.find(G);
    if (it != k->xrec.end()) { *out = &it->second; return BAMM_OK; }
    const uint32_t maskY = (uint32_t)(ipow4(K + 1) - 1);
    ExcK::XRec x;
    x.h_B.resize(s->n);                                      // every element is written by the loop below
    x.h_lo.resize(s->n);
    RawVec<uint4> xrec(s->n);
    auto stream_y = [&](uint64_t n, int64_t pos) -> uint32_t {      // kmer_ mod 4^(K+1) as the stream alone gives it
        uint32_t y = 0;
        for (uint32_t d = 0; d <= K; d++) {
            const int64_t q = pos - (int64_t)d;
            if (q < 0) break;                                        // implicit A-padding (Sequence.cpp:35-41)
            const uint32_t w = s->h_words[s->h_word_off[n] + (uint64_t)(q >> 4)];
            y |= ((w >> (30u - 2u * (uint32_t)(q & 15))) & 3u) << (2u * d);
        }
        return y;
    };
    host_ranges(s->n, [&](uint64_t n_begin, uint64_t n_end) {
    for (uint64_t n = n_begin; n < n_end; n++) {
        const uint64_t e0 = k->h_off[n], e1 = k->h_off[n + 1];
        x.h_B[n] = 0; x.h_lo[n] = 0; xrec[n] = make_uint4(0, 0, 0, 0);
        if (e0 == e1) continue;
        const uint32_t lo = k->h_ex[e0].x, hi = k->h_ex[e1 - 1].x, L = s->h_len[n];
        const uint32_t hiB = std::min(hi + G - 1u, L - 1u);
        const uint32_t B = hiB - lo + 1u;
        // the record holds 12 seven-bit y fields (Y <= 64), or 9 ten-bit ones at K = 3 (Y = 256)
        const uint32_t max_fields = K == 3u ? 9u : 12u;
        if (B > 8u || B + G - 1u > max_fields || lo >= 4096u) { x.h_B[n] = 255; continue; }
        x.h_B[n] = (uint8_t)B;
        x.h_lo[n] = lo;
        uint32_t w3[3] = {0, 0, 0};
        uint64_t e = e0;
        for (uint32_t i = 0; i < B + G - 1u; i++) {
            const int64_t pos = (int64_t)lo - (int64_t)(G - 1u) + i;
            uint32_t y = maskY + 1u;                                 // no such position
            if (pos >= 0) {
                while (e < e1 && (int64_t)k->h_ex[e].x < pos) e++;
                y = (e < e1 && (int64_t)k->h_ex[e].x == pos) ? k->h_ex[e].y : stream_y(n, pos);
            }
            // the fields form ONE bit string over the three words, 7 bits each (10 at K = 3), field i at bit i * width:
            // a fix lane's consecutive fields are a single funnel shift of two neighbouring words (grouped_kernel.h: xrec_fields)
            const uint32_t bit = (K == 3u ? 10u : 7u) * i, wd = bit >> 5, sh = bit & 31u;
            w3[wd] |= y << sh;
            if (sh + (K == 3u ? 10u : 7u) > 32u) w3[wd + 1u] |= y >> (32u - sh);
        }
        xrec[n] = make_uint4(lo | (B << 12), w3[0], w3[1], w3[2]);
    }
    });
    int rc = dev_upload(s->ctx, &x.d_xrec, xrec.data(), xrec.size());
    if (rc) return rc;
    if (hipStreamSynchronize(s->ctx->stream) != hipSuccess) {
        (void)hipFree(x.d_xrec);
        set_error("stream sync failed while uploading the sequence records");
        return BAMM_ERR_HIP;
    }
    auto ins = k->xrec.emplace(G, std::move(x));
    *out = &ins.first->second;
    return BAMM_OK;
}

uint32_t default_threads(const bamm_ctx* c, int mclass) {
    uint32_t t = c->threads ? c->threads : max_threads_for_mclass(mclass);
    return std::min(t, max_threads_for_mclass(mclass));
}

// block size of one launch: the grouped kernel's longer length classes are built for fewer waves
uint32_t bucket_threads(const bamm_ctx* c, const EmBucket& b) {
    if (b.mclass == kLongClass) return 256u;
    const uint32_t t = default_threads(c, b.mclass);
    return b.grouped ? std::min(t, grp_max_threads(kMClasses[b.mclass])) : t;
}

uint32_t default_blocks(const bamm_ctx* c, uint32_t threads) {
    if (c->blocks) return c->blocks;
    const uint32_t cus = c->num_cus > 0 ? (uint32_t)c->num_cus : 256u;
    return cus * std::max(1u, 2048u / threads);       // fill 32 waves per CU
}

// a process may drive several devices (one context each): make the context's device current
int use_device(const bamm_ctx* c) {
    BAMM_HIP(hipSetDevice(c->device));
    return BAMM_OK;
}

// An event pair costs 7-8 us of stream time per pass on gfx950 (0.9 % of a 1M-sequence iteration, 6.5 % of a 125k-sequence
// one: profiles/r04_timing_every_cost.txt).  Three ways: every `timing_every`-th pass of a call bracketed; none; or
// (BAMM_TIMING_WHOLE_CALL) one pair around ALL passes of a call -- every pass covered, nothing between two passes, the
// launch gaps inside the interval.
static int event_pair(bamm_em* em) {
    if (em->events_used == em->events.size()) {
        hipEvent_t a, b;
        BAMM_HIP(hipEventCreate(&a));
        BAMM_HIP(hipEventCreate(&b));
        em->events.emplace_back(a, b);
        em->event_passes.push_back(1u);
    }
    return BAMM_OK;
}
int record_event(bamm_em* em, bool start) {
    if (em->timing_every == BAMM_TIMING_WHOLE_CALL) {
        if (!start) return BAMM_OK;
        em->pass_no++;
        if (!em->region_open) {
            if (int rc = event_pair(em)) return rc;
            BAMM_HIP(hipEventRecord(em->events[em->events_used].first, em->ctx->stream));
            em->region_open = true;
            em->region_passes = 0;
        }
        em->region_passes++;
        return BAMM_OK;
    }
    if (start) {
        em->timing_now = em->timing_every != 0 && em->pass_no % em->timing_every == 0;
        em->pass_no++;
    }
    if (!em->timing_now) return BAMM_OK;
    if (start) {
        if (int rc = event_pair(em)) return rc;
        BAMM_HIP(hipEventRecord(em->events[em->events_used].first, em->ctx->stream));
    } else {
        BAMM_HIP(hipEventRecord(em->events[em->events_used].second, em->ctx->stream));
        em->event_passes[em->events_used] = 1u;
        em->events_used++;
    }
    return BAMM_OK;
}
// the second event of a whole-call interval: behind the last pass's kernels
int close_timed_region(bamm_em* em) {
    if (!em->region_open) return BAMM_OK;
    em->region_open = false;
    BAMM_HIP(hipEventRecord(em->events[em->events_used].second, em->ctx->stream));
    em->event_passes[em->events_used] = em->region_passes;
    em->events_used++;
    return BAMM_OK;
}
struct TimedRegionCloser { bamm_em* em; ~TimedRegionCloser() { (void)close_timed_region(em); } };

// one bucket through the fused kernel of its flavour (grouped columns or one column at a time)
int launch_fused(bamm_em* em, const EmBucket& eb, bool accum, bool write_r, EmKernelArgs& a, uint32_t threads,
                 hipStream_t st, const UpdateArgs* fuse = nullptr) {
    if (!eb.grouped) {
        a.logC = eb.logc;
        a.sparse_cap = accum ? eb.sparse_cap : 0u;
        a.sparse_wave_bytes = accum ? eb.sparse_bytes : 0u;
        return launch_em_seq(eb.mclass, accum, write_r, a, eb.blocks, threads, st);
    }
    GrpKernelArgs ga{};
    a.logC = eb.logc;
    a.sparse_cap = 0; a.sparse_wave_bytes = 0;
    ga.e = a;
    ga.xrec = eb.d_xrec;
    if (!grp_geometry(em->prm.K, em->prm.W, eb.G, kMClasses[eb.mclass], threads / 64u, accum, accum ? eb.logc : 0u, eb.layout, &ga.g)) {
        set_error("grouped kernel geometry does not fit (K=%u W=%u)", em->prm.K, em->prm.W);
        return BAMM_ERR_UNSUPPORTED;
    }
    if ((em->prm.K == 3u || (eb.layout & 8u)) && accum) {      // kernels whose fix lanes log their sums
        const size_t waves = (size_t)eb.blocks * (threads / 64u);
        const size_t cap = ((eb.count + waves - 1) / waves) * std::min<size_t>(64, (size_t)ga.g.Bv * ga.g.T);   // entries per wave
        const size_t need = waves * cap;                      // 8-byte entries
        if (need > em->fix_log_words) {
            if (em->d_fix_log) { scratch_free(em->ctx, em->d_fix_log); em->d_fix_log = nullptr; em->fix_log_words = 0; }
            if (int rc = scratch_alloc(em->ctx, &em->d_fix_log, need)) return rc;
            em->fix_log_words = need;
        }
        ga.fix_log = em->d_fix_log;
        ga.fix_log_cap = (uint32_t)cap;
    }
    if (fuse) {                                              // the previous pass's update runs in this launch's prologue
        ga.fused = 1u; ga.upd = *fuse; ga.upd_off = em->fuse_upd_off; ga.s_block = em->d_s_block;
    }
    if (em->peer_on && accum && !write_r) {                  // in-kernel all-reduce in the launch's tail (the pass's only launch)
        em->pass_summed_in_kernel = true;
        comm_peer_args(em->comm, &ga.peer);
        ga.peer.words = (uint32_t)em->cells + 3u;
        ga.peer.ticket = em->d_peer_words; ga.peer.err = em->d_peer_words + 1;
        ga.peer.timeout_ticks = (unsigned long long)em->ctx->peer_timeout_ms * 100000ull;      // 100 MHz wall clock
        ga.peer.seq = comm_peer_next_seq(em->comm);          // the ranks count their passes in step
        ga.peer.slot = (uint32_t)(ga.peer.seq % 3ull);
    }
    return launch_em_grp(eb.mclass, accum, write_r, ga, eb.blocks, threads, st);
}

// Make the kernel(s) one bucket's passes launch ready ahead of the first pass: the HIP runtime loads a translation unit's
// code object on the first use of any kernel in it (0.8 ms for the mixed-row kernels, 8-14 ms for the 4 MB units), which
// otherwise lands in the handle's first pass.  Runs on a thread of its own beside bamm_em_create's host work.
int prime_bucket(bamm_ctx* c, const bamm_em_params& prm, uint32_t Y, bool sliced, const EmBucket& eb) {
    if (hipSetDevice(c->device) != hipSuccess) { set_error("hipSetDevice failed"); return BAMM_ERR_HIP; }
    EmKernelArgs a{};
    a.K = prm.K; a.W = prm.W; a.Y = Y;
    if (eb.mclass == kLongClass) return launch_long_em(a, true, false, false, kPrimeOnly, c->stream);
    const uint32_t threads = bucket_threads(c, eb);
    if (sliced || !eb.grouped) {                             // k_em_seq, the slices and the list walk share kernels.hip
        a.logC = 0;
        return launch_em_seq(eb.mclass, false, false, a, kPrimeOnly, threads, c->stream);
    }
    GrpKernelArgs ga{};
    ga.e = a;
    if (!grp_geometry(prm.K, prm.W, eb.G, kMClasses[eb.mclass], threads / 64u, true, eb.logc, eb.layout, &ga.g)) return BAMM_OK;   // (the launch reports it)
    return launch_em_grp(eb.mclass, true, false, ga, kPrimeOnly, threads, c->stream);
}

void prepare_update(bamm_em* em, bool q_window, bool fused, UpdateArgs& u);

EmBook capture_book(const bamm_em* em) {
    return EmBook{em->d_s, em->d_s_alt, em->d_q, em->d_v, em->d_v_alt, em->s_last, em->q_last, em->d_acc, em->acc_cur, em->llh_cur,
                  em->host_iteration, em->events_used, em->pass_no, em->estep_done, em->acc_dirty, em->mask_done, em->ring_prev_dirty};
}
void restore_book(bamm_em* em, const EmBook& b) {
    em->d_s = b.d_s; em->d_s_alt = b.d_s_alt; em->d_q = b.d_q; em->d_v = b.d_v; em->d_v_alt = b.d_v_alt;
    em->s_last = b.s_last; em->q_last = b.q_last; em->d_acc = b.d_acc; em->acc_cur = b.acc_cur; em->llh_cur = b.llh_cur;
    em->host_iteration = b.host_iteration; em->events_used = b.events_used; em->pass_no = b.pass_no;
    em->estep_done = b.estep_done; em->acc_dirty = b.acc_dirty; em->mask_done = b.mask_done; em->ring_prev_dirty = b.ring_prev_dirty;
}

// clear whatever a pass left unconsumed (accumulate without update, getR replay, a fused sequence cut short)
int clean_accumulator(bamm_em* em) {
    if (!em->acc_dirty) return BAMM_OK;
    hipStream_t st = em->ctx->stream;
    if (em->d_acc_ring) {
        BAMM_HIP(hipMemsetAsync(em->d_acc_ring, 0, 3 * em->acc_stride * sizeof(long long), st));
        em->acc_cur = 0; em->d_acc = em->d_acc_ring; em->ring_prev_dirty = false;
    } else {
        BAMM_HIP(hipMemsetAsync(em->d_acc, 0, (em->cells + 3) * sizeof(long long), st));
    }
    em->acc_dirty = false;
    return BAMM_OK;
}

// local E(+M) pass over every length bucket; every block adds its table into the pass's accumulator.
// fuse_q_window >= 0: the PREVIOUS pass's model update (with that q-window flag) runs in the block prologue of this
// pass's first launch instead of a k_update launch of its own (em->fusable handles, accumulating passes only).
int run_accumulate(bamm_em* em, bool accum, bool replay_last = false, bool dense_r = false, int fuse_q_window = -1) {
    bamm_seqs* s = em->seqs;
    hipStream_t st = em->ctx->stream;
    int rc = use_device(em->ctx);
    if (rc) return rc;
    UpdateArgs fuse{};
    em->pass_summed_in_kernel = false;
    const bool fusing = fuse_q_window >= 0;
    if (fusing) prepare_update(em, fuse_q_window != 0, true, fuse);     // consumes the previous pass's sums on the stream
    else if ((rc = clean_accumulator(em))) return rc;
    if (em->d_nnz && accum)                                   // sliced path: this pass's count of non-zero windows starts at 0
        BAMM_HIP(hipMemsetAsync(em->d_nnz + (em->nnz_prev_slot ^ 1u), 0, sizeof(unsigned long long), st));
    if ((rc = record_event(em, true))) return rc;
    for (size_t b = 0; b < em->ebuckets.size(); b++) {
        const EmBucket& bk = em->ebuckets[b];
        EmKernelArgs a{};
        a.sv = make_view(s, em->exc, bk, em->d_mask);
        a.K = em->prm.K; a.W = em->prm.W; a.Y = em->Y;
        a.logC = bk.logc;
        a.s = replay_last ? em->s_last : em->d_s;
        a.q = replay_last ? em->q_last : em->d_q;
        a.acc = em->d_acc;
        a.fix_scale = ldexpf(1.0f, (int)em->fix_shift - 40);
        a.stop = em->stop_arg;
        a.r_out = nullptr; a.r_base = 0; a.seq_begin = 0; a.seq_end = 0;
        if (bk.mclass == kLongClass) {
            // the sliced path's getR() reads dense r from d_state (slot layout unless the E pass is k_em_seq)
            const bool want_r = em->sliced && dense_r;
            if (want_r && !em->d_state && (rc = scratch_alloc(em->ctx, &em->d_state, (size_t)s->total_len))) return rc;
            a.r_out = em->d_state;
            if ((rc = launch_long_em(a, accum, want_r, want_r && !em->e_fused, bk.blocks, st))) return rc;
            continue;
        }
        const uint32_t threads = bucket_threads(em->ctx, bk);
        if (!em->sliced) {
            rc = launch_fused(em, bk, accum, false, a, threads, st, (fusing && b == 0) ? &fuse : nullptr);
        } else {
            uint32_t widest = 0;
            for (auto& sl : em->m_slices) widest = std::max(widest, sl.second - sl.first);
            // compacted lists instead of dense r between the E pass and the M slices: when the whole odds table is
            // in LDS (the E pass is k_em_seq) and a wave's copy of the decoded sequence fits beside the count slice
            const bool lists = em->e_fused && em->d_list_r && !dense_r &&
                               m_list_lds_bytes(widest, em->Y, em->m_slice_logc, kMClasses[bk.mclass], threads / 64u) <= 160u * 1024u;
            // ... and per pass, decided on the device: while the model is uninformative every window has a non-zero
            // addend and the list walk costs far more than the dense one (config 4, pass 1: 26.5 against 15.4 ms,
            // profiles/r03_c4_cold_passes.txt).  Both flavours of the pass are enqueued; the count the previous pass's
            // E kernel took picks the one that runs (the other's launches return at entry).
            const bool adaptive = lists && accum && em->d_nnz && em->adaptive_lists;
            if ((!lists || adaptive) && !em->d_state) {
                if ((rc = scratch_alloc(em->ctx, &em->d_state, (size_t)s->total_len))) return rc;
            }
            auto flavour = [&](bool use_lists, int run_if_long) -> int {   // run_if_long: -1 = unconditional
                EmKernelArgs f = a;
                int r = BAMM_OK;
                f.r_out = em->d_state;
                if (run_if_long >= 0) {
                    f.nnz_prev = em->d_nnz + em->nnz_prev_slot; f.nnz_limit = em->nnz_limit; f.run_if_long = (uint32_t)run_if_long;
                }
                if (em->d_nnz && accum) f.nnz_out = em->d_nnz + (em->nnz_prev_slot ^ 1u);
                if (em->e_fused) {
                    // the whole odds table fits LDS (only the count table does not): the fused kernel's E
                    // pass, leaving r in the reference's layout (k_em_seq WRITE_R) or the lists
                    f.seq_end = (uint32_t)s->n;
                    f.logC = 0; f.sparse_cap = 0; f.sparse_wave_bytes = 0;
                    if (use_lists) { f.list_r = em->d_list_r; f.list_p = em->d_list_p; f.list_n = em->d_list_n; }
                    r = launch_em_seq(bk.mclass, false, true, f, bk.blocks, threads, st);
                    if (use_lists) {
                        f.logC = em->m_slice_logc;
                        for (size_t i = 0; accum && i < em->m_slices.size() && !r; i++)
                            r = launch_m_list(bk.mclass, f, em->m_slices[i].first, em->m_slices[i].second, bk.blocks, threads, st);
                        return r;
                    }
                } else {
                    for (size_t i = 0; i < em->e_slices.size() && !r; i++)
                        r = launch_e_slice(bk.mclass, f, em->e_slices[i].first, em->e_slices[i].second,
                                           i + 1 == em->e_slices.size(), bk.blocks, threads, st);
                }
                f.logC = em->m_slice_logc;
                {   // this bucket's list capacity: what fits next to the widest slice's count table
                    const size_t table = m_slice_lds_bytes(widest, em->Y, em->m_slice_logc);
                    uint32_t cap = em->m_slice_cap;
                    while (cap && table + (threads / 64u) * m_slice_wave_bytes(kMClasses[bk.mclass], cap) > 160u * 1024u) cap -= 64u;
                    f.sparse_cap = cap;
                    f.sparse_wave_bytes = (uint32_t)m_slice_wave_bytes(kMClasses[bk.mclass], cap);
                }
                for (size_t i = 0; accum && i < em->m_slices.size() && !r; i++)
                    r = launch_m_slice(bk.mclass, f, em->m_slices[i].first, em->m_slices[i].second, em->e_fused,
                                       bk.blocks, threads, st);
                return r;
            };
            if (adaptive) {
                rc = flavour(false, 1);                      // dense r while the lists would be long
                if (!rc) rc = flavour(true, 0);
            } else {
                rc = flavour(lists, -1);
            }
        }
        if (rc) return rc;
    }
    rc = record_event(em, false);
    if (rc) return rc;
    if (!replay_last) { em->s_last = em->d_s; em->q_last = em->d_q; em->mask_done = false; }
    if (em->d_nnz && accum) em->nnz_prev_slot ^= 1u;          // the next pass chooses from what this one counted
    em->acc_dirty = true;                                     // until the update (or the E-only read-out) has consumed it
    return BAMM_OK;
}

// int64 sum of `n_words` words across the ranks, on the context's stream: RCCL or the caller's callback
int allreduce_words(bamm_em* em, void* dev_ptr, size_t n_words) {
    if (em->comm) return comm_allreduce_i64(em->comm, dev_ptr, n_words, em->ctx->stream);
    if (em->allreduce && em->allreduce(em->allreduce_user, dev_ptr, n_words, (void*)em->ctx->stream) != 0) {
        set_error("all-reduce callback failed");
        return BAMM_ERR_COMM;
    }
    return BAMM_OK;
}

int run_allreduce(bamm_em* em) {
    // mode 2: a pass whose launch carried the tail left the all-reduced sums in the accumulator (launch_fused,
    // peer_allreduce_tail).  Every other pass of such a handle -- EStep() alone (an E-only launch has no tail), EM::mask's
    // kernels, a replay -- is summed by the communicator's collective like in mode 1: same integers either way.
    if (em->pass_summed_in_kernel) { em->pass_summed_in_kernel = false; return BAMM_OK; }
    if (em->comm) return comm_allreduce_i64(em->comm, em->d_acc, em->cells + 3, em->ctx->stream);
    if (!em->allreduce) return BAMM_OK;
    int rc = em->allreduce(em->allreduce_user, em->d_acc, em->cells + 3, (void*)em->ctx->stream);
    if (rc != 0) {
        set_error("all-reduce callback failed with %d", rc);
        return BAMM_ERR_COMM;
    }
    return BAMM_OK;
}

// The slot a new q may be written to: never the one the last E pass read (q_last: getR() and the
// MStep() replay recompute r from it, the reference's r_ keeps the EStep's q, EM.cpp:139-200), the
// current slot when that is free, else the third one.
float* q_write_slot(bamm_em* em) {
    if (em->d_q != em->q_last) return em->d_q;
    for (float* p : em->d_qbuf)
        if (p != em->q_last) return p;
    return em->d_q;
}

// Fill the arguments of one model update and move the host's bookkeeping past it (the launch that carries it --
// k_update, or the next pass's first sequence kernel when `fused` -- follows on the stream).
// q_window: this pass is one of the first five of its optimize() / iterate() call, where the reference
// re-estimates q (`iteration` is local to EM::optimize, EM.cpp:75-99)
void prepare_update(bamm_em* em, bool q_window, bool fused, UpdateArgs& u) {
    u = UpdateArgs{};
    u.K = em->prm.K; u.W = em->prm.W; u.Kbg = em->Kbg;
    u.acc = em->d_acc; u.count_unit = ldexp(1.0, -(int)em->fix_shift); u.vbg = em->d_vbg; u.A = em->d_A; u.n = em->d_n; u.s = em->d_s_alt;
    float* q_out = q_write_slot(em);
    if (fused)                                               // every block reads d_q while the writer block stores q_out: never the same slot
        for (float* p : em->d_qbuf)
            if (p != em->q_last && p != em->d_q) { q_out = p; break; }
    u.q = em->d_q; u.q_out = q_out; u.status = em->d_status; u.trace = em->d_trace; u.trace_cap = em->prm.max_iterations;
    u.iteration = em->d_iteration; u.optimize_q = (em->prm.optimize_q && q_window) ? 1 : 0;
    u.n_seqs_override = (double)em->prm.n_seqs_global;
    u.llh_in = em->d_llh[em->llh_cur]; u.llh_out = em->d_llh[em->llh_cur ^ 1u];
    u.partial = em->d_upd_partial; u.ticket = em->d_upd_ticket;
    if (em->stop_arg) {
        u.stop = em->d_stop; u.epsilon = em->prm.epsilon; u.opt_iteration = em->opt_iteration;
        u.llh_prev = em->opt_llh_prev; u.llh_prev_from_status = em->opt_iteration > 1u ? 1 : 0;
        u.status_mirror = em->d_status_mirror ? em->d_status_mirror + 8 * (em->opt_iteration & 1u) : nullptr;
    }
    if (fused) {
        // every block of the carrying launch reads slot `acc_cur` and the old v; its writer block stores the new v
        // elsewhere and clears the slot after next; the launch's own pass adds into the next slot
        u.v_old = em->d_v; u.v = em->d_v_alt;
        u.acc_zero = em->d_acc_ring + (size_t)((em->acc_cur + 2u) % 3u) * em->acc_stride;
        std::swap(em->d_v, em->d_v_alt);
        em->acc_cur = (em->acc_cur + 1u) % 3u;
        em->d_acc = em->d_acc_ring + (size_t)em->acc_cur * em->acc_stride;
        em->ring_prev_dirty = true;                          // the slot just read stays as it is until the next update clears it
    } else {
        u.v = em->d_v; u.v_old = nullptr;
        u.acc_zero = em->ring_prev_dirty ? em->d_acc_ring + (size_t)((em->acc_cur + 2u) % 3u) * em->acc_stride : nullptr;
        em->ring_prev_dirty = false;
    }
    em->acc_dirty = false;                                    // consumed (k_update zeroes it; the ring moves on)
    std::swap(em->d_s, em->d_s_alt);
    em->d_q = q_out;
    em->llh_cur ^= 1u;
    em->host_iteration++;
    em->estep_done = false;
    em->books[em->host_iteration & 3u] = capture_book(em);
}

int run_update(bamm_em* em, bool q_window) {
    int rc = use_device(em->ctx);
    if (rc) return rc;
    UpdateArgs u;
    prepare_update(em, q_window, false, u);
    return launch_update(u, em->ctx->stream);
}

// A collective that was already enqueued when a peer aborted the communicator completes with whatever it had: what
// was computed from it is not a model.  Every read-out that follows a stream synchronisation says so.
int comm_still_sound(const bamm_em* em) {
    if (em->comm && comm_aborted(em->comm)) {
        set_error("the communicator was aborted while passes were in flight: the handle's model is not valid");
        return BAMM_ERR_COMM;
    }
    if (em->peer_on && em->d_peer_words) {                   // the stream is idle: did a block give up waiting for a peer?
        uint32_t w[2] = {0, 0};
        BAMM_HIP(hipMemcpy(w, em->d_peer_words, sizeof w, hipMemcpyDeviceToHost));
        if (w[1] != 0u) {
            set_error("in-kernel all-reduce: the sums of rank %u did not arrive within the deadline (peer_timeout_ms); the handle's model is not valid", w[1] - 1u);
            return BAMM_ERR_COMM;
        }
    }
    return BAMM_OK;
}

int fetch_status(bamm_em* em) {
    BAMM_HIP(hipSetDevice(em->ctx->device));
    BAMM_HIP(hipMemcpyAsync(em->h_status, em->d_status, 8 * sizeof(float), hipMemcpyDeviceToHost, em->ctx->stream));
    BAMM_HIP(hipStreamSynchronize(em->ctx->stream));
    return comm_still_sound(em);
}

// In front of the first pass that all-reduces over a communicator (every rank is in that call, on a thread or a
// process of its own): the ranks sum their sequence counts and compare their accumulator units.  A unit too fine for
// the SUM (8 ranks of 2 M sequences at 2^-40: a cell could reach 2^64) or units that differ would wrap or mis-scale
// silently; here they are an error on every rank at once.
int verify_comm(bamm_em* em) {
    if (!em->comm || em->comm_verified) return BAMM_OK;
    uint32_t world = 1;
    (void)bamm_comm_info(em->comm, nullptr, &world, nullptr);
    long long want_peer = 0;                                 // ranks whose context asks for the in-kernel all-reduce
    if (world > 1) {
        long long h[4] = {(long long)em->seqs->n, (long long)em->fix_shift, (long long)em->fix_shift * (long long)em->fix_shift,
                          em->ctx->use_peer_allreduce ? 1 : 0};
        int rc = use_device(em->ctx);
        if (!rc && !em->d_comm_words) rc = dev_alloc(&em->d_comm_words, 4);
        if (rc) return rc;
        long long* const d = em->d_comm_words;
        hipStream_t st = em->ctx->stream;
        hipError_t e = hipMemcpyAsync(d, h, sizeof h, hipMemcpyHostToDevice, st);
        if (e == hipSuccess) rc = comm_allreduce_i64(em->comm, d, 4, st);
        if (e == hipSuccess && !rc) e = hipMemcpyAsync(h, d, sizeof h, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess && !rc) e = hipStreamSynchronize(st);
        if (rc) return rc;
        if (e != hipSuccess) { set_error("accumulator-unit check over the communicator: %s", hipGetErrorString(e)); return BAMM_ERR_HIP; }
        want_peer = h[3];
        if ((long long)world * h[2] != h[1] * h[1]) {
            set_error("the ranks' accumulator units differ (bamm_em_params.n_seqs_bound must be the same on every rank)");
            return BAMM_ERR_ARG;
        }
        uint32_t bits = 0;
        while ((uint64_t(1) << bits) < (uint64_t)h[0] && bits < 63u) bits++;
        if (em->fix_shift > std::min(40u, 62u - std::min(bits, 38u))) {
            set_error("%lld sequences over %u ranks need a coarser accumulator unit than 2^-%u: pass their number as "
                      "bamm_em_params.n_seqs_bound on every rank", h[0], world, em->fix_shift);
            return BAMM_ERR_ARG;
        }
    }
    // in-kernel all-reduce, when the context asks for it: every rank must be able to (a pass of ONE launch of the mixed-row
    // kernel, which is built with the tail) and must have mapped every peer's inbox -- the ranks vote, a single refusal keeps
    // all of them on RCCL
    em->peer_on = false;
    if (world > 1 && want_peer == (long long)world) {        // (asked for on every rank: the set-up below is a collective)
        constexpr uint32_t kStride = 2056;                   // entries per (slot, source): 2048 top-order cells + 3 statistics, padded
        // (the tail is built into k_em_mix -- K = 2, both strands, the widths the planner gives mixed rows: the bench / config
        // 2 / 3 / 5 shapes -- and into k_em_grp's classes up to BAMM_FUSE_MAX_M positions per lane at K <= 2: single strand, k = 0 / 1,
        // other widths)
        const EmBucket& eb0 = em->ebuckets[0];
        const bool can = em->ebuckets.size() == 1u && eb0.grouped && eb0.mclass != kLongClass &&
                         ((eb0.layout & 8u) != 0u || (em->prm.K <= 2u && kMClasses[eb0.mclass] <= BAMM_FUSE_MAX_M)) &&
                         world <= kPeerMaxWorld && em->cells + 3u <= kStride && !em->allreduce;
        int ready = 0;
        // (every rank goes through the set-up, able or not: it is a collective; the vote inside it counts mapped inboxes)
        int rc = comm_peer_setup(em->comm, kStride, &ready);
        if (rc) return rc;
        long long vote[1] = {(ready && can) ? 1 : 0};
        if ((rc = use_device(em->ctx))) return rc;
        long long* const d = em->d_comm_words;               // (allocated above: world > 1)
        hipStream_t st = em->ctx->stream;
        hipError_t e = hipMemcpyAsync(d, vote, sizeof vote, hipMemcpyHostToDevice, st);
        if (e == hipSuccess) rc = comm_allreduce_i64(em->comm, d, 1, st);
        if (e == hipSuccess && !rc) e = hipMemcpyAsync(vote, d, sizeof vote, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess && !rc) e = hipStreamSynchronize(st);
        if (rc) return rc;
        if (e != hipSuccess) { set_error("in-kernel all-reduce vote: %s", hipGetErrorString(e)); return BAMM_ERR_HIP; }
        if (vote[0] == (long long)world) {
            if (!em->d_peer_words) {
                if ((rc = dev_alloc(&em->d_peer_words, 2))) return rc;
                BAMM_HIP(hipMemsetAsync(em->d_peer_words, 0, 2 * sizeof(uint32_t), st));
            }
            em->peer_on = true;
            em->peer_note.clear();
        } else {
            em->peer_note = !ready ? std::string("inboxes: ") + comm_peer_why(em->comm)
                          : !can ? "this handle's pass is not one launch of a grouped-column kernel built with the tail (K <= 2, one length class of at most 1024 positions, no N-rich sequences beside it)"
                                 : "another rank could not";
        }
    } else if (em->ctx->use_peer_allreduce) {
        em->peer_note = world > 1 ? "not every rank's context asked for it" : "one rank: nothing to reduce";
    }
    em->comm_verified = true;
    return BAMM_OK;
}

}  // namespace

namespace bamm {
int ctx_device(const bamm_ctx* c) { return c->device; }
hipStream_t ctx_stream(const bamm_ctx* c) { return c->stream; }
}  // namespace bamm

extern "C" {

// ------------------------------------------------------------------------------ context ----
int bamm_ctx_create(int device, void* hip_stream, bamm_ctx** out) {
    if (!out) { set_error("bamm_ctx_create: null out"); return BAMM_ERR_ARG; }
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
        set_error("no HIP device visible: the gfx950 extension cannot run (there is no CPU fallback)");
        return BAMM_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= count) {
        set_error("device %d out of range (0..%d)", device, count - 1);
        return BAMM_ERR_ARG;
    }
    BAMM_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    BAMM_HIP(hipGetDeviceProperties(&prop, device));
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0) {
        set_error("device %d is %s; this library only carries gfx950 code objects", device, prop.gcnArchName);
        return BAMM_ERR_NO_DEVICE;
    }
    bamm_ctx* c = new bamm_ctx();
    c->device = device;
    c->num_cus = prop.multiProcessorCount;
    c->name = prop.name;
    if (c->name.empty()) c->name = prop.gcnArchName;       // some driver stacks leave the marketing name blank
    if (hip_stream) {
        c->stream = (hipStream_t)hip_stream;
    } else {
        hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        if (e != hipSuccess) {
            set_error("hipStreamCreate failed: %s", hipGetErrorString(e));
            delete c;
            return BAMM_ERR_HIP;
        }
        c->own_stream = true;
    }
    (void)prime_model_kernels();                             // k_make_s / k_update: the first kernels any handle launches
    {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) c->scratch_cap_bytes = total_b / 4;
        std::lock_guard<std::mutex> g(g_ctx_mu);
        g_ctxs.push_back(c);
    }
    *out = c;
    return BAMM_OK;
}

int bamm_device_count(int* n) {
    if (!n) { set_error("bamm_device_count: null argument"); return BAMM_ERR_ARG; }
    *n = 0;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
        set_error("no HIP device visible");
        return BAMM_ERR_NO_DEVICE;
    }
    *n = count;
    return BAMM_OK;
}

int bamm_device_pci_bus_id(int device, char* buf, size_t cap) {
    if (!buf || cap < 16) { set_error("bamm_device_pci_bus_id: buffer of at least 16 bytes"); return BAMM_ERR_ARG; }
    buf[0] = 0;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) { set_error("no HIP device visible"); return BAMM_ERR_NO_DEVICE; }
    if (device < 0 || device >= count) { set_error("device %d out of range (0..%d)", device, count - 1); return BAMM_ERR_ARG; }
    BAMM_HIP(hipDeviceGetPCIBusId(buf, (int)cap, device));
    return BAMM_OK;
}

int bamm_device_can_access_peer(int device, int peer, int* can) {
    if (!can) { set_error("bamm_device_can_access_peer: null argument"); return BAMM_ERR_ARG; }
    *can = 0;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) { set_error("no HIP device visible"); return BAMM_ERR_NO_DEVICE; }
    if (device < 0 || device >= count || peer < 0 || peer >= count) { set_error("device %d / peer %d out of range (0..%d)", device, peer, count - 1); return BAMM_ERR_ARG; }
    if (device == peer) { *can = 1; return BAMM_OK; }
    BAMM_HIP(hipDeviceCanAccessPeer(can, device, peer));
    return BAMM_OK;
}

int bamm_ctx_destroy(bamm_ctx* c) {
    if (!c) return BAMM_OK;
    {
        std::lock_guard<std::mutex> g(g_ctx_mu);
        g_ctxs.erase(std::remove(g_ctxs.begin(), g_ctxs.end(), c), g_ctxs.end());
    }
    (void)hipSetDevice(c->device);
    if (!c->scratch_idle.empty()) (void)hipStreamSynchronize(c->stream);
    for (auto& b : c->scratch_idle) (void)hipFree(b.first);
    if (c->stage_buf[0]) {
        (void)hipStreamSynchronize(c->stream);
        (void)hipHostFree(c->stage_buf[0]);
        for (hipEvent_t e : c->stage_ev) if (e) (void)hipEventDestroy(e);
    }
    if (c->own_stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return BAMM_OK;
}

int bamm_ctx_sync(bamm_ctx* c) {
    if (!c) { set_error("null ctx"); return BAMM_ERR_ARG; }
    BAMM_HIP(hipStreamSynchronize(c->stream));
    return BAMM_OK;
}

int bamm_ctx_device_name(bamm_ctx* c, char* buf, size_t cap) {
    if (!c || !buf || !cap) { set_error("bad argument"); return BAMM_ERR_ARG; }
    snprintf(buf, cap, "%s", c->name.c_str());
    return BAMM_OK;
}

int bamm_ctx_set_launch(bamm_ctx* c, uint32_t blocks, uint32_t threads) {
    if (!c || (threads & 63u) || threads > 1024u) { set_error("threads must be a multiple of 64 <= 1024"); return BAMM_ERR_ARG; }
    c->blocks = blocks;
    c->threads = threads;
    return BAMM_OK;
}

int bamm_ctx_set_tuning(bamm_ctx* c, const char* key, int value) {
    if (!c || !key) { set_error("bamm_ctx_set_tuning: null argument"); return BAMM_ERR_ARG; }
    const std::string k(key);
    if (k == "grouped") c->use_grouped = value != 0;
    else if (k == "sparse") c->use_sparse = value != 0;
    else if (k == "e_fused") c->use_e_fused = value != 0;
    else if (k == "e_list") c->use_e_list = value != 0;
    else if (k == "fused_update") c->use_fused_update = value != 0;
    else if (k == "adaptive_lists") c->use_adaptive_lists = value != 0;
    else if (k == "update_blocks") c->use_update_blocks = value != 0;
    else if (k == "scratch_poison") c->scratch_poison = value != 0;
    else if (k == "peer_allreduce") c->use_peer_allreduce = value != 0;
    else if (k == "peer_timeout_ms") {
        if (value < 1 || value > 600000) { set_error("peer_timeout_ms must be 1..600000"); return BAMM_ERR_ARG; }
        c->peer_timeout_ms = (uint32_t)value;
    }
    else if (k == "scratch_cache_mb") {
        if (value < 0) { set_error("scratch_cache_mb must be >= 0"); return BAMM_ERR_ARG; }
        c->scratch_cap_bytes = (size_t)value << 20;
        if (value == 0) (void)flush_idle_scratch(c->device);
    }
    else if (k == "list_threshold_pct") {
        if (value < 0 || value > 100) { set_error("list_threshold_pct must be 0..100"); return BAMM_ERR_ARG; }
        c->list_threshold_pct = (uint32_t)value;
    }
    else if (k == "group_size") {
        if (value != 0 && (value < 2 || value > 4)) { set_error("group_size must be 0 (auto) or 2..4"); return BAMM_ERR_ARG; }
        c->group_size = (uint32_t)value;
    } else if (k == "group_layout") {
        if (value < -1 || (value > 3 && value != 8)) { set_error("group_layout must be -1 (auto), 0..3 or 8 (mixed rows)"); return BAMM_ERR_ARG; }
        c->group_layout = value;
    } else { set_error("bamm_ctx_set_tuning: unknown key '%s'", key); return BAMM_ERR_ARG; }
    return BAMM_OK;
}

// ------------------------------------------------------------------------------ sequences --
// device arrays of a whole packed set that its maker already holds (bamm_seqs_from_codes): the resident set takes them over
// instead of uploading the host copies again.  A pointer the set has taken is nulled here; the caller frees what is left.
struct AdoptDev { uint32_t* words; uint64_t* word_off; uint32_t* len; uint64_t* pos_off; };   // words: 80 words of slack behind the stream

static int seqs_upload_impl(bamm_ctx* c, const bamm_packed* p, uint64_t begin, uint64_t end, bamm_seqs** out, AdoptDev* have) {
    if (!c || !p || !out || begin > end || end > p->n_seqs) {
        set_error("bamm_seqs_upload: bad argument");
        return BAMM_ERR_ARG;
    }
    if (end - begin > 0xfffffff0ull) { set_error("more than 2^32 sequences per device"); return BAMM_ERR_UNSUPPORTED; }
    BAMM_HIP(hipSetDevice(c->device));
    std::unique_ptr<bamm_seqs> s(new bamm_seqs());
    s->ctx = c;
    s->n = end - begin;
    const uint64_t w0 = p->word_off[begin], w1 = p->word_off[end];
    std::vector<uint64_t> woff(s->n + 1);
    s->h_len.assign(p->len + begin, p->len + end);
    s->h_pos_off.resize(s->n + 1);
    s->h_exc_off.resize(s->n + 1);
    const uint64_t e0 = p->exc_off[begin], e1 = p->exc_off[end];
    uint64_t pos = 0;
    s->min_len = s->n ? UINT32_MAX : 0;
    for (uint64_t n = 0; n < s->n; n++) {
        woff[n] = p->word_off[begin + n] - w0;
        s->h_pos_off[n] = pos;
        s->h_exc_off[n] = p->exc_off[begin + n] - e0;
        pos += s->h_len[n];
        s->max_len = std::max(s->max_len, s->h_len[n]);
        s->min_len = std::min(s->min_len, s->h_len[n]);
    }
    woff[s->n] = w1 - w0;
    s->h_words.resize(w1 - w0); par_memcpy(s->h_words.data(), p->words + w0, (w1 - w0) * sizeof(uint32_t));
    s->h_word_off = woff;
    s->h_pos_off[s->n] = pos;
    s->h_exc_off[s->n] = e1 - e0;
    s->total_len = pos;
    s->h_exc_pos.resize(e1 - e0); par_memcpy(s->h_exc_pos.data(), p->exc_pos + e0, (e1 - e0) * sizeof(uint32_t));
    s->h_exc_kmer.resize(e1 - e0); par_memcpy(s->h_exc_kmer.data(), p->exc_kmer + e0, (e1 - e0) * sizeof(uint32_t));
    s->h_exc_clean.resize(e1 - e0); par_memcpy(s->h_exc_clean.data(), p->exc_clean + e0, (e1 - e0) * sizeof(uint32_t));

    // length buckets: one kernel instantiation per positions-per-lane class
    // (+ one bucket for the sequences beyond the longest class: long_seq.hip walks those window by window)
    std::vector<std::vector<uint32_t>> members(kNumMClasses + 1);
    for (uint64_t n = 0; n < s->n; n++) {
        const int mc = m_class_for_len(s->h_len[n]);
        members[mc < 0 ? kNumMClasses : mc].push_back((uint32_t)n);
    }
    int rc;
    // 80 zero words of slack: the grouped kernel reads a lane's words without checking the sequence's end
    if (have && begin == 0 && end == p->n_seqs) {
        s->d_words = have->words; s->d_word_off = have->word_off; s->d_len = have->len; s->d_pos_off = have->pos_off;
        have->words = nullptr; have->word_off = nullptr; have->len = nullptr; have->pos_off = nullptr;   // the set's from here on
        BAMM_HIP(hipMemsetAsync(s->d_words + (w1 - w0), 0, 80 * sizeof(uint32_t), c->stream));
    } else {
        if ((rc = dev_alloc(&s->d_words, (w1 - w0) + 80))) return rc;
        BAMM_HIP(hipMemsetAsync(s->d_words + (w1 - w0), 0, 80 * sizeof(uint32_t), c->stream));
        if ((rc = ctx_upload(c, s->d_words, p->words + w0, (w1 - w0) * sizeof(uint32_t)))) return rc;
        if ((rc = dev_upload(c, &s->d_word_off, woff.data(), woff.size()))) return rc;
        if ((rc = dev_upload(c, &s->d_len, s->h_len.data(), s->h_len.size()))) return rc;
        if ((rc = dev_upload(c, &s->d_pos_off, s->h_pos_off.data(), s->h_pos_off.size()))) return rc;
    }
    int used = 0;
    for (int mc = 0; mc <= kNumMClasses; mc++) used += !members[mc].empty();
    for (int mc = 0; mc <= kNumMClasses; mc++) {
        if (members[mc].empty()) continue;
        Bucket b;
        b.mclass = mc < kNumMClasses ? mc : kLongClass;
        b.count = (uint32_t)members[mc].size();
        if (mc < kNumMClasses) b.work = (double)b.count * kMClasses[mc];
        else for (uint32_t n : members[mc]) b.work += s->h_len[n] / 8.0;   // ~8x the cost per position of the fast kernels
        s->buckets.push_back(b);
        if (used > 1) {
            if ((rc = dev_upload(c, &s->buckets.back().d_idx, members[mc].data(), members[mc].size()))) return rc;
            s->buckets.back().h_idx = std::move(members[mc]);
        }
    }
    BAMM_HIP(hipStreamSynchronize(c->stream));
    s->hbm_bytes = (w1 - w0) * 4 + (s->n + 1) * 8 * 2 + s->n * 4;
    *out = s.release();
    return BAMM_OK;
}

int bamm_seqs_upload(bamm_ctx* c, const bamm_packed* p, uint64_t begin, uint64_t end, bamm_seqs** out) {
    return seqs_upload_impl(c, p, begin, end, out, nullptr);
}

// bamm_pack_codes_seeded on the device (csrc/prep.hip), then bamm_seqs_upload: the same packed set, the same resident set
int bamm_seqs_from_codes(bamm_ctx* c, const uint8_t* codes, const uint64_t* off, uint64_t n_seqs, int single_strand, uint32_t seed,
                         bamm_packed** packed_out, bamm_seqs** seqs_out) {
    if (!c || !packed_out || (n_seqs && (!codes || !off))) { set_error("bamm_seqs_from_codes: null argument"); return BAMM_ERR_ARG; }
    *packed_out = nullptr;
    if (seqs_out) *seqs_out = nullptr;
    if (n_seqs == 0) {
        int rc0 = bamm_pack_codes_seeded(codes, off, 0, single_strand, seed, packed_out);
        if (!rc0 && seqs_out) rc0 = bamm_seqs_upload(c, *packed_out, 0, 0, seqs_out);
        return rc0;
    }
    for (uint64_t n = 0; n < n_seqs; n++) {
        const uint64_t L0 = off[n + 1] - off[n];
        if ((single_strand ? L0 : 2 * L0 + 1) > 0xffffffffull) { set_error("sequence %llu longer than 2^32-1", (unsigned long long)n); return BAMM_ERR_ARG; }
    }
    BAMM_HIP(hipSetDevice(c->device));
    hipStream_t st = c->stream;
    const uint64_t n_codes = off[n_seqs] - off[0];
    std::vector<void*> owned;                                // everything allocated here is freed on every path out
    auto release = [&](int rc) { for (void* p : owned) (void)hipFree(p); return rc; };
    auto alloc = [&](auto** p, size_t count) -> int {
        int rc = dev_alloc(p, count ? count : 1);
        if (!rc) owned.push_back((void*)*p);
        return rc;
    };
    int rc;
    uint8_t* d_codes = nullptr;
    uint64_t* d_off = nullptr;
    PrepArgs a{};
    if ((rc = alloc(&d_codes, n_codes)) || (rc = alloc(&d_off, n_seqs + 1))) return release(rc);
    {
        // the records as one contiguous run starting at 0 (off[0] may be anything)
        std::vector<uint64_t> rel(n_seqs + 1);
        for (uint64_t n = 0; n <= n_seqs; n++) rel[n] = off[n] - off[0];
        if (ctx_upload(c, d_codes, codes + off[0], n_codes) != BAMM_OK || ctx_upload(c, d_off, rel.data(), (n_seqs + 1) * sizeof(uint64_t)) != BAMM_OK ||
            hipStreamSynchronize(st) != hipSuccess) { set_error("bamm_seqs_from_codes: upload failed"); return release(BAMM_ERR_HIP); }
    }
    a.codes = d_codes; a.off = d_off; a.n = n_seqs; a.single_strand = single_strand;
    if ((rc = alloc(&a.len, n_seqs)) || (rc = alloc(&a.word_off, n_seqs + 1)) || (rc = alloc(&a.pos_off, n_seqs + 1)) ||
        (rc = alloc(&a.zero_off, n_seqs + 1)) || (rc = alloc(&a.draw_off, n_seqs + 1)) || (rc = alloc(&a.exc_off, n_seqs + 1))) return release(rc);
    for (uint64_t* p : {a.word_off, a.pos_off, a.zero_off, a.draw_off, a.exc_off})
        if (hipMemsetAsync(p, 0, sizeof(uint64_t), st) != hipSuccess) { set_error("hipMemsetAsync failed"); return release(BAMM_ERR_HIP); }
    if ((rc = launch_prep_count(a, st))) return release(rc);
    for (uint64_t* p : {a.word_off, a.pos_off, a.zero_off, a.draw_off})
        if ((rc = launch_scan_u64(p, n_seqs + 1, st))) return release(rc);
    uint64_t tot[4] = {0, 0, 0, 0};                          // words, positions, zeros, draws
    {
        uint64_t* src[4] = {a.word_off, a.pos_off, a.zero_off, a.draw_off};
        for (int i = 0; i < 4; i++)
            if (hipMemcpyAsync(&tot[i], src[i] + n_seqs, sizeof(uint64_t), hipMemcpyDeviceToHost, st) != hipSuccess) { set_error("read-back failed"); return release(BAMM_ERR_HIP); }
        if (hipStreamSynchronize(st) != hipSuccess) { set_error("bamm_seqs_from_codes: the counting pass failed"); return release(BAMM_ERR_HIP); }
    }
    uint8_t* d_draws = nullptr;
    if ((rc = alloc(&a.zero_pos, tot[2])) || (rc = alloc(&d_draws, tot[3]))) return release(rc);
    if ((rc = launch_prep_zeros(a, st))) return release(rc);
    {
        // the one serial resource of Sequence::Sequence is libc's rand() stream: the draws are taken on the host (all
        // threads enter the stream by jump-ahead, pack.cpp) while the device lists the zero positions
        std::vector<uint8_t> draws(tot[3] ? tot[3] : 1);
        rand_draws_mod4(seed, tot[3], draws.data());
        if (tot[3] && (ctx_upload(c, d_draws, draws.data(), tot[3]) != BAMM_OK || hipStreamSynchronize(st) != hipSuccess)) {
            set_error("bamm_seqs_from_codes: upload of the draws failed"); return release(BAMM_ERR_HIP);
        }
    }
    a.draws = d_draws;
    if ((rc = launch_prep_pack(a, false, st)) || (rc = launch_scan_u64(a.exc_off, n_seqs + 1, st))) return release(rc);
    uint64_t n_exc = 0;
    if (hipMemcpyAsync(&n_exc, a.exc_off + n_seqs, sizeof(uint64_t), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
        set_error("bamm_seqs_from_codes: the exception count failed"); return release(BAMM_ERR_HIP);
    }
    if ((rc = alloc(&a.words, tot[0] + 80)) || (rc = alloc(&a.exc_pos, n_exc)) || (rc = alloc(&a.exc_kmer, n_exc)) || (rc = alloc(&a.exc_clean, n_exc))) return release(rc);
    if ((rc = launch_prep_pack(a, true, st))) return release(rc);
    // the host's view of the packed set (malloc: bamm_packed_free releases it)
    bamm_packed* p = (bamm_packed*)calloc(1, sizeof(bamm_packed));
    if (!p) { set_error("out of memory"); return release(BAMM_ERR_ARG); }
    p->n_seqs = n_seqs; p->n_words = tot[0]; p->n_exc = n_exc; p->total_len = tot[1];
    p->words = (uint32_t*)malloc((tot[0] ? tot[0] : 1) * sizeof(uint32_t));
    p->word_off = (uint64_t*)malloc((n_seqs + 1) * sizeof(uint64_t));
    p->len = (uint32_t*)malloc(n_seqs * sizeof(uint32_t));
    p->exc_off = (uint64_t*)malloc((n_seqs + 1) * sizeof(uint64_t));
    p->exc_pos = (uint32_t*)calloc(n_exc ? n_exc : 1, sizeof(uint32_t));
    p->exc_kmer = (uint32_t*)calloc(n_exc ? n_exc : 1, sizeof(uint32_t));
    p->exc_clean = (uint32_t*)calloc(n_exc ? n_exc : 1, sizeof(uint32_t));
    bool ok = p->words && p->word_off && p->len && p->exc_off && p->exc_pos && p->exc_kmer && p->exc_clean;
    auto down = [&](void* dst, const void* src, size_t bytes) { if (ok && bytes) ok = ctx_download(c, dst, src, bytes) == BAMM_OK; };
    down(p->words, a.words, tot[0] * sizeof(uint32_t));
    down(p->word_off, a.word_off, (n_seqs + 1) * sizeof(uint64_t));
    down(p->len, a.len, n_seqs * sizeof(uint32_t));
    down(p->exc_off, a.exc_off, (n_seqs + 1) * sizeof(uint64_t));
    down(p->exc_pos, a.exc_pos, n_exc * sizeof(uint32_t));
    down(p->exc_kmer, a.exc_kmer, n_exc * sizeof(uint32_t));
    down(p->exc_clean, a.exc_clean, n_exc * sizeof(uint32_t));
    if (ok) ok = hipStreamSynchronize(st) == hipSuccess;
    if (!ok) { bamm_packed_free(p); set_error("bamm_seqs_from_codes: the packed set could not be brought back"); return release(BAMM_ERR_HIP); }
    uint32_t mx = 0, mn = UINT32_MAX;
    for (uint64_t n = 0; n < n_seqs; n++) { mx = std::max(mx, p->len[n]); mn = std::min(mn, p->len[n]); }
    p->max_len = mx; p->min_len = mn;
    if (seqs_out) {
        // the stream, its offsets and the lengths are on the device already: the resident set takes those arrays over
        AdoptDev have{a.words, a.word_off, a.len, a.pos_off};
        rc = seqs_upload_impl(c, p, 0, n_seqs, seqs_out, &have);
        for (void* taken : {(void*)a.words, (void*)a.word_off, (void*)a.len, (void*)a.pos_off}) {
            const bool left = taken == (void*)have.words || taken == (void*)have.word_off || taken == (void*)have.len || taken == (void*)have.pos_off;
            if (!left) owned.erase(std::remove(owned.begin(), owned.end(), taken), owned.end());
        }
        if (rc) { bamm_packed_free(p); return release(rc); }
    }
    release(BAMM_OK);
    *packed_out = p;
    return BAMM_OK;
}

// BackgroundModel's counting pass over a resident set (BackgroundModel.cpp:26-42), calculateV on the host (:441-473)
int bamm_seqs_bg_model(bamm_ctx* c, bamm_seqs* s, uint32_t K, const float* alpha, float* vbg_out) {
    if (!c || !s || !alpha || !vbg_out || K > BAMM_MAX_ORDER) { set_error("bamm_seqs_bg_model: bad argument"); return BAMM_ERR_ARG; }
    if (s->ctx != c) { set_error("sequence set belongs to another context"); return BAMM_ERR_ARG; }
    const size_t Y = ipow4(K + 1);
    std::vector<uint64_t> top(Y, 0);
    if (s->n) {
        BAMM_HIP(hipSetDevice(c->device));
        ExcK* exc = nullptr;
        int rc = exceptions_for_order(s, K, &exc);
        if (rc) return rc;
        unsigned long long* d_counts = nullptr;
        if ((rc = dev_alloc(&d_counts, Y))) return rc;
        hipError_t e = hipMemsetAsync(d_counts, 0, Y * sizeof(unsigned long long), c->stream);
        if (e == hipSuccess) rc = launch_bg_counts(s->d_words, s->d_word_off, s->d_len, exc->d_off, exc->d_exc, s->n, K, d_counts,
                                                   (uint32_t)std::max(1, c->num_cus), c->stream);
        if (e == hipSuccess && !rc) e = hipMemcpyAsync(top.data(), d_counts, Y * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess && !rc) e = hipStreamSynchronize(c->stream);
        (void)hipFree(d_counts);
        if (rc) return rc;
        if (e != hipSuccess) { set_error("bamm_seqs_bg_model: %s", hipGetErrorString(e)); return BAMM_ERR_HIP; }
    }
    bg_from_top_counts(top.data(), K, alpha, vbg_out);
    return BAMM_OK;
}

// SeqGenerator::sample_bgseqset_by_fold (SeqGenerator.cpp:63-348) on the device: csrc/negs.hip
int bamm_sample_negatives(bamm_ctx* c, bamm_seqs* pos, uint32_t s_order, uint64_t m_fold, int generic, uint64_t keep_stride,
                          bamm_packed** packed_out, bamm_seqs** seqs_out) {
    if (!c || !pos || !packed_out || m_fold == 0) { set_error("bamm_sample_negatives: bad argument"); return BAMM_ERR_ARG; }
    *packed_out = nullptr;
    if (seqs_out) *seqs_out = nullptr;
    if (pos->ctx != c) { set_error("sequence set belongs to another context"); return BAMM_ERR_ARG; }
    if (s_order != kNegMaxOrder) {
        set_error("the device sampler is written for -s 2 (SeqGenerator.cpp:112-186); other orders run on the host");
        return BAMM_ERR_UNSUPPORTED;
    }
    if (pos->max_len > BAMM_MAX_SEQ_POSITIONS || pos->n == 0) { set_error("the device sampler takes non-empty sets of sequences up to %u positions", BAMM_MAX_SEQ_POSITIONS); return BAMM_ERR_UNSUPPORTED; }
    if (!GlibcRandStream::libc_is_this_generator()) {
        set_error("libc's rand() is not the restated glibc generator on this host: the sampler runs on the host, drawing from libc itself");
        return BAMM_ERR_UNSUPPORTED;
    }
    BAMM_HIP(hipSetDevice(c->device));
    hipStream_t st = c->stream;
    ExcK* exc = nullptr;
    int rc = exceptions_for_order(pos, s_order, &exc);
    if (rc) return rc;
    std::vector<void*> owned;
    auto release = [&](int r) { for (void* p : owned) (void)hipFree(p); return r; };
    auto alloc = [&](auto** p, size_t count) -> int { int r = dev_alloc(p, count ? count : 1); if (!r) owned.push_back((void*)*p); return r; };
    NegArgs a{};
    a.words = pos->d_words; a.word_off = pos->d_word_off; a.len = pos->d_len; a.exc_off = exc->d_off; a.exc = exc->d_exc;
    a.n = pos->n; a.s = s_order; a.generic = generic; a.m_fold = m_fold; a.keep_stride = keep_stride;
    for (uint32_t k = 0; k <= kNegMaxOrder; k++) a.A[k] = 20.0f;            // SeqGenerator.cpp:29-32
    const uint32_t tot = (uint32_t)bg_size(s_order);
    float *d_v = nullptr, *d_bar = nullptr;
    if ((rc = alloc(&a.total_counts, tot)) || (rc = alloc(&d_v, tot)) || (rc = alloc(&d_bar, tot)) || (rc = alloc(&a.bad, 1))) return release(rc);
    if (hipMemsetAsync(a.total_counts, 0, tot * sizeof(unsigned long long), st) != hipSuccess || hipMemsetAsync(a.bad, 0, sizeof(uint32_t), st) != hipSuccess) {
        set_error("hipMemsetAsync failed"); return release(BAMM_ERR_HIP);
    }
    if ((rc = launch_neg_counts(a, st))) return release(rc);
    std::vector<unsigned long long> cnt(tot);
    if (hipMemcpyAsync(cnt.data(), a.total_counts, tot * sizeof(unsigned long long), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
        set_error("bamm_sample_negatives: the counting pass failed"); return release(BAMM_ERR_HIP);
    }
    // the set's conditionals and their bars from the totals (SeqGenerator.cpp:86-110, the float order of host/fdr.cpp)
    std::vector<float> v(tot, 0.f), bar(tot, 0.f);
    {
        auto off = [](uint32_t k) { return (uint32_t)bg_offset(k); };
        unsigned long long norm = 0;
        for (uint32_t y = 0; y < 4; y++) norm += cnt[y];
        float sum = 0.0f;
        for (uint32_t y = 0; y < 4; y++) {
            v[y] = ((float)cnt[y] + a.A[0] * 0.25f) / ((float)norm + a.A[0]);
            sum += v[y];
            bar[y] = sum;
        }
        for (uint32_t k = 1; k <= s_order; k++) {
            sum = 0.f;
            for (uint32_t y = 0; y < (uint32_t)ipow4(k + 1); y++) {
                const uint32_t yk = y / 4, y2 = y % (uint32_t)ipow4(k);
                v[off(k) + y] = ((float)cnt[off(k) + y] + a.A[k] * v[off(k - 1) + y2]) / ((float)cnt[off(k - 1) + yk] + a.A[k]);
                if (y % 4 == 0) sum = 0.f;
                sum += v[off(k) + y];
                bar[off(k) + y] = sum;
            }
        }
    }
    // where every positive's draws start, where its kept negatives go; the generator's seed state and the powers t^(2^b)
    const uint64_t total_neg = pos->n * m_fold;
    auto kept = [&](uint64_t idx) { return keep_stride <= 1 || (idx % keep_stride == 0 && idx + keep_stride <= total_neg); };
    std::vector<uint64_t> draw0(pos->n + 1, 0), wo(pos->n + 1, 0), first_kept(pos->n + 1, 0);
    for (uint64_t i = 0; i < pos->n; i++) {
        const uint64_t L = pos->h_len[i];
        uint64_t nk = 0;
        if (keep_stride <= 1) nk = m_fold;
        else for (uint64_t f = 0; f < m_fold; f++) nk += kept(i * m_fold + f);
        draw0[i + 1] = draw0[i] + L * m_fold;
        wo[i + 1] = wo[i] + nk * ((L + 15) / 16);
        first_kept[i + 1] = first_kept[i] + nk;
    }
    const uint64_t n_neg = first_kept[pos->n], n_words = wo[pos->n];
    GlibcRandStream g;
    g.seed(42u);                                             // SeqGenerator.cpp:35
    std::vector<uint32_t> pw(48 * 31, 0);
    {
        uint32_t base[31] = {0, 1}, tmp[31];
        for (int b = 0; b < 48; b++) {
            memcpy(pw.data() + 31 * b, base, sizeof base);
            GlibcRandStream::poly_mul(base, base, tmp);
            memcpy(base, tmp, sizeof tmp);
        }
    }
    uint64_t *d_draw0 = nullptr, *d_wo = nullptr;
    uint32_t *d_seed = nullptr, *d_pw = nullptr;
    if ((rc = alloc(&d_draw0, pos->n)) || (rc = alloc(&d_wo, pos->n)) || (rc = alloc(&d_seed, 34)) || (rc = alloc(&d_pw, pw.size())) ||
        (rc = alloc(&a.out_words, n_words))) return release(rc);
    hipError_t e = hipMemcpyAsync(d_v, v.data(), tot * sizeof(float), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(d_bar, bar.data(), tot * sizeof(float), hipMemcpyHostToDevice, st);
    if (e == hipSuccess && (ctx_upload(c, d_draw0, draw0.data(), pos->n * sizeof(uint64_t)) || ctx_upload(c, d_wo, wo.data(), pos->n * sizeof(uint64_t)))) e = hipErrorUnknown;
    if (e == hipSuccess) e = hipMemcpyAsync(d_seed, g.r, 34 * sizeof(uint32_t), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(d_pw, pw.data(), pw.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st);
    if (e != hipSuccess) { set_error("bamm_sample_negatives: upload failed: %s", hipGetErrorString(e)); return release(BAMM_ERR_HIP); }
    a.v = d_v; a.bar = d_bar; a.draw0 = d_draw0; a.out_word_off = d_wo; a.seed_state = d_seed; a.pow2 = d_pw;
    if ((rc = launch_neg_sample(a, st))) return release(rc);
    // the negatives as a packed set of their own (single strand, no unknown base: no exceptions)
    bamm_packed* p = (bamm_packed*)calloc(1, sizeof(bamm_packed));
    if (!p) { set_error("out of memory"); return release(BAMM_ERR_ARG); }
    p->n_seqs = n_neg; p->n_words = n_words;
    p->words = (uint32_t*)malloc((n_words ? n_words : 1) * sizeof(uint32_t));
    p->word_off = (uint64_t*)calloc(n_neg + 1, sizeof(uint64_t));
    p->len = (uint32_t*)calloc(n_neg ? n_neg : 1, sizeof(uint32_t));
    p->exc_off = (uint64_t*)calloc(n_neg + 1, sizeof(uint64_t));
    p->exc_pos = (uint32_t*)calloc(1, sizeof(uint32_t));
    p->exc_kmer = (uint32_t*)calloc(1, sizeof(uint32_t));
    p->exc_clean = (uint32_t*)calloc(1, sizeof(uint32_t));
    uint32_t bad = 0;
    bool ok = p->words && p->word_off && p->len && p->exc_off && p->exc_pos && p->exc_kmer && p->exc_clean;
    if (ok && n_words) ok = ctx_download(c, p->words, a.out_words, n_words * sizeof(uint32_t)) == BAMM_OK;
    if (ok) ok = hipMemcpyAsync(&bad, a.bad, sizeof bad, hipMemcpyDeviceToHost, st) == hipSuccess;
    if (ok) ok = hipStreamSynchronize(st) == hipSuccess;
    if (!ok) { bamm_packed_free(p); set_error("bamm_sample_negatives: the sampling pass failed"); return release(BAMM_ERR_HIP); }
    release(BAMM_OK);
    if (bad) {                                               // rand() == RAND_MAX at a first base: the reference leaves that byte unset
        bamm_packed_free(p);
        set_error("a first base drew rand() == RAND_MAX, which the reference leaves undefined: sample this set on the host");
        return BAMM_ERR_UNSUPPORTED;
    }
    uint64_t at = 0, total = 0;
    uint32_t mx = 0, mn = UINT32_MAX;
    for (uint64_t i = 0; i < pos->n; i++) {
        const uint32_t L = pos->h_len[i];
        for (uint64_t k = first_kept[i]; k < first_kept[i + 1]; k++) {
            p->len[k] = L; p->word_off[k] = at; at += (L + 15) / 16; total += L;
            mx = std::max(mx, L); mn = std::min(mn, L);
        }
    }
    p->word_off[n_neg] = at;
    p->total_len = total; p->max_len = mx; p->min_len = n_neg ? mn : 0;
    *packed_out = p;
    if (seqs_out && (rc = bamm_seqs_upload(c, p, 0, n_neg, seqs_out))) { bamm_packed_free(p); *packed_out = nullptr; return rc; }
    return BAMM_OK;
}

int bamm_seqs_destroy(bamm_seqs* s) {
    if (!s) return BAMM_OK;
    {
        std::lock_guard<std::mutex> lock(s->mu);
        if (--s->refs > 0) return BAMM_OK;
    }
    (void)hipSetDevice(s->ctx->device);
    delete s;
    return BAMM_OK;
}

int bamm_seqs_info(const bamm_seqs* s, uint64_t* n_seqs, uint64_t* total_len, uint32_t* max_len, uint64_t* hbm_bytes) {
    if (!s) { set_error("null seqs"); return BAMM_ERR_ARG; }
    if (n_seqs) *n_seqs = s->n;
    if (total_len) *total_len = s->total_len;
    if (max_len) *max_len = s->max_len;
    if (hbm_bytes) *hbm_bytes = s->hbm_bytes;
    return BAMM_OK;
}

// ------------------------------------------------------------------------------ EM ---------
int bamm_em_destroy(bamm_em* em) {
    if (!em) return BAMM_OK;
    (void)hipSetDevice(em->ctx->device);
    (void)hipStreamSynchronize(em->ctx->stream);
    for (void* p : {(void*)em->d_vbg, (void*)em->d_A, (void*)em->d_v, (void*)em->d_n, (void*)em->d_s, (void*)em->d_qbuf[0],
                    (void*)em->d_status, (void*)em->d_trace, (void*)em->d_iteration, (void*)em->d_mask, (void*)em->d_acc_ring,
                    (void*)em->d_v_alt, (void*)em->d_llh[0], (void*)em->d_s_block, (void*)em->d_nnz, (void*)em->d_upd_partial, (void*)em->d_upd_ticket,
                    (void*)em->d_state, (void*)em->d_list_r, (void*)em->d_list_p, (void*)em->d_list_n, (void*)em->d_s_alt, (void*)em->d_fix_log,
                    (void*)em->d_qbuf[1], (void*)em->d_qbuf[2],
                    (void*)em->d_mask_r, (void*)em->d_mask_bits, (void*)em->d_mask_hist, (void*)em->d_mask_sel, (void*)em->d_mask_qseq,
                    (void*)em->d_mask_partial_n, (void*)em->d_mask_partial_stat})
        scratch_free(em->ctx, p);                             // set-sized blocks go back to the context, the rest is freed
    for (uint32_t* p : em->owned_idx) (void)hipFree(p);
    if (em->h_status) (void)hipHostFree(em->h_status);
    (void)hipFree(em->d_stop);
    (void)hipFree(em->d_peer_words);
    (void)hipFree(em->d_comm_words);
    for (hipEvent_t e : em->opt_events) if (e) (void)hipEventDestroy(e);
    for (auto& ev : em->events) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
    bamm_seqs_destroy(em->seqs);
    delete em;
    return BAMM_OK;
}

int bamm_em_create(bamm_ctx* c, bamm_seqs* seqs, const bamm_em_params* prm, const float* vbg, const float* A,
                   const float* v_init, const uint8_t* seq_mask, bamm_em** out) {
    if (!c || !seqs || !prm || !vbg || !A || !v_init || !out) { set_error("bamm_em_create: null argument"); return BAMM_ERR_ARG; }
    *out = nullptr;
    if (seqs->ctx != c) { set_error("sequence set belongs to another context"); return BAMM_ERR_ARG; }
    if (prm->K > BAMM_MAX_ORDER) { set_error("order %u > %d (kmer_ spans 11 bases)", prm->K, BAMM_MAX_ORDER); return BAMM_ERR_ARG; }
    if (prm->W == 0) { set_error("motif width 0"); return BAMM_ERR_ARG; }
    if (seqs->n && seqs->min_len < prm->W) {
        set_error("a sequence of length %u is shorter than the motif (W=%u); the reference drops those before EM (mainBaMM.cpp:75-83)",
                  seqs->min_len, prm->W);
        return BAMM_ERR_ARG;
    }
    const uint32_t Y = (uint32_t)ipow4(prm->K + 1);
    const size_t kLds = 160 * 1024;
    bool sliced = em_lds_bytes(prm->W, Y, true, 0, 0) > kLds;
    uint32_t e_cols = 0, m_cols = 0;
    // orders 7..10 (kmer_ spans 11 bases, Sequence.cpp:37): not even one column of the odds / count tables (4^(K+1) rows)
    // fits the 160 KiB of a CU.  Those models run with their tables in global memory (long_seq.hip: every window
    // multiplies its W odds straight from the table, the fixed-point addends go straight into the pass's
    // accumulator) -- the same integers, written for coverage, not speed.
    bool global_tables = false;
    if (sliced) {
        while (e_cols < prm->W && e_slice_lds_bytes(e_cols + 1, Y) <= kLds) e_cols++;
        while (m_cols < prm->W && m_slice_lds_bytes(m_cols + 1, Y, 0) <= kLds) m_cols++;
        if (e_cols == 0 || m_cols == 0) { global_tables = true; sliced = false; }
    }
    BAMM_HIP(hipSetDevice(c->device));
    bamm_em* em = new bamm_em();
    em->ctx = c;
    em->seqs = seqs;
    { std::lock_guard<std::mutex> lock(seqs->mu); seqs->refs++; }
    em->prm = *prm;
    if (em->prm.max_iterations == 0) em->prm.max_iterations = 1000;
    em->Y = Y;
    em->Kbg = std::min(prm->bg_order, prm->K);           // EM.cpp:23
    em->vsz = v_size(prm->K, prm->W);
    em->cells = (size_t)Y * prm->W;
    em->sliced = sliced;
    if (sliced) {
        auto cut = [&](uint32_t max_cols, std::vector<std::pair<uint32_t, uint32_t>>& out) {
            const uint32_t n = (prm->W + max_cols - 1) / max_cols, per = (prm->W + n - 1) / n;
            for (uint32_t j = 0; j < prm->W; j += per) out.emplace_back(j, std::min(prm->W, j + per));
            return per;
        };
        cut(e_cols, em->e_slices);
        em->e_fused = em_lds_bytes(prm->W, Y, false, 0, 0) <= kLds && c->use_e_fused;
        // sparse M-slices: room for a list of 256 windows + the y of every position per wave (8 waves per
        // block at the longest length class) is taken off the column budget when that costs no extra slice
        int mc_max = 0;                                      // longest length class present (the long bucket has no class)
        for (auto& b : seqs->buckets) mc_max = std::max(mc_max, b.mclass);
        const int Mmax = kMClasses[mc_max];
        const uint32_t waves = max_threads_for_mclass(mc_max) / 64u;
        const size_t scratch = !c->use_sparse ? 0 : m_slice_wave_bytes(Mmax, 256) * waves;
        uint32_t m_cols_sparse = 0;
        while (scratch && m_cols_sparse < prm->W && m_slice_lds_bytes(m_cols_sparse + 1, Y, 0) + scratch <= kLds) m_cols_sparse++;
        // ... and when it would, the widest slices stay and every bucket takes the longest list that still
        // fits beside them (run_accumulate): at k=4, W=30 a 128-entry list next to the 120 KB count slice
        // is worth 20 % of the iteration
        const bool roomy = m_cols_sparse && (prm->W + m_cols_sparse - 1) / m_cols_sparse == (prm->W + m_cols - 1) / m_cols;
        if (roomy) m_cols = m_cols_sparse;
        if (scratch) em->m_slice_cap = 256;
        const uint32_t per_m = cut(m_cols, em->m_slices);
        const size_t used = roomy ? scratch : std::min(scratch, kLds - std::min(kLds, m_slice_lds_bytes(per_m, Y, 0)));
        while (em->m_slice_logc < 4 && m_slice_lds_bytes(per_m, Y, em->m_slice_logc + 1) + used <= kLds) em->m_slice_logc++;
    }
    hipStream_t st = c->stream;
    int rc = BAMM_OK;
    // the code objects of the kernels the handle will launch are loaded beside the host work below (prime_bucket), each as
    // soon as the plan names the kernel; joined on every way out
    struct Primers { std::vector<std::thread> t; ~Primers() { for (auto& x : t) if (x.joinable()) x.join(); } } primers;
    auto prime = [&primers, c, prm_copy = em->prm, Y, sliced](const EmBucket& eb) {
        primers.t.emplace_back([c, prm_copy, Y, sliced, eb] { (void)prime_bucket(c, prm_copy, Y, sliced, eb); });
    };
    auto fail = [&](int code) { bamm_em_destroy(em); return code; };
    if ((rc = exceptions_for_order(seqs, prm->K, &em->exc))) return fail(rc);
    if ((rc = dev_upload(c, &em->d_vbg, vbg, bg_size(prm->bg_order)))) return fail(rc);
    if ((rc = dev_upload(c, &em->d_A, A, (size_t)(prm->K + 1) * prm->W))) return fail(rc);
    if ((rc = dev_upload(c, &em->d_v, v_init, em->vsz))) return fail(rc);
    if ((rc = dev_alloc(&em->d_n, em->vsz))) return fail(rc);
    if ((rc = dev_alloc(&em->d_s, (size_t)prm->W * (Y + 1)))) return fail(rc);
    if ((rc = dev_alloc(&em->d_s_alt, (size_t)prm->W * (Y + 1)))) return fail(rc);
    for (auto& slot : em->d_qbuf)
        if ((rc = dev_upload(c, &slot, &prm->q, 1))) return fail(rc);
    em->d_q = em->d_qbuf[0];
    if ((rc = dev_alloc(&em->d_status, 8))) return fail(rc);
    if ((rc = dev_alloc(&em->d_trace, (size_t)em->prm.max_iterations * 3))) return fail(rc);
    if ((rc = dev_alloc(&em->d_iteration, 1))) return fail(rc);
    em->acc_stride = (em->cells + 3 + 1) & ~(size_t)1;      // slots start on 16-byte boundaries
    if ((rc = dev_alloc(&em->d_acc_ring, 3 * em->acc_stride))) return fail(rc);
    em->d_acc = em->d_acc_ring;
    if ((rc = dev_alloc(&em->d_v_alt, em->vsz))) return fail(rc);
    if ((rc = dev_alloc(&em->d_llh[0], 2))) return fail(rc);
    em->d_llh[1] = em->d_llh[0] + 1;
    if (!update_fits_lds(prm->K, prm->W) && c->use_update_blocks) {
        if ((rc = dev_alloc(&em->d_upd_partial, kUpdateMaxBlocks)) || (rc = dev_alloc(&em->d_upd_ticket, 1))) return fail(rc);
        if (hipMemsetAsync(em->d_upd_ticket, 0, sizeof(uint32_t), st) != hipSuccess) { set_error("hipMemsetAsync failed"); return fail(BAMM_ERR_HIP); }
    }
    {   // counts are sums of r * 2^fix_shift over at most n_seqs_global (else this handle's) sequences, each
        // contributing less than 1 per cell: keep the int64 total below 2^62
        const uint64_t n_hint = std::max<uint64_t>(prm->n_seqs_bound ? prm->n_seqs_bound : (prm->n_seqs_global ? prm->n_seqs_global : seqs->n), 1);
        uint32_t bits = 0;
        while ((uint64_t(1) << bits) < n_hint && bits < 63u) bits++;
        em->fix_shift = std::min(40u, 62u - std::min(bits, 38u));
    }
    if (hipMemsetAsync(em->d_n, 0, em->vsz * sizeof(float), st) != hipSuccess ||
        hipMemsetAsync(em->d_status, 0, 8 * sizeof(float), st) != hipSuccess ||
        hipMemsetAsync(em->d_iteration, 0, sizeof(uint32_t), st) != hipSuccess ||
        hipMemsetAsync(em->d_llh[0], 0, 2 * sizeof(float), st) != hipSuccess ||
        hipMemsetAsync(em->d_acc_ring, 0, 3 * em->acc_stride * sizeof(long long), st) != hipSuccess) {
        set_error("hipMemsetAsync failed");
        return fail(BAMM_ERR_HIP);
    }
    if (seq_mask && seqs->n)
        if ((rc = dev_upload(c, &em->d_mask, seq_mask, seqs->n))) return fail(rc);
    em->n_active = seqs->n;
    if (seq_mask) em->n_active = (uint64_t)std::count_if(seq_mask, seq_mask + seqs->n, [](uint8_t m) { return m != 0; });
    if (hipHostMalloc((void**)&em->h_status, 24 * sizeof(float) + 16 * sizeof(unsigned long long), hipHostMallocDefault) != hipSuccess) {
        set_error("hipHostMalloc failed");
        return fail(BAMM_ERR_HIP);
    }
    memset(em->h_status, 0, 24 * sizeof(float) + 16 * sizeof(unsigned long long));
    em->h_tagged = reinterpret_cast<unsigned long long*>(em->h_status + 24);
    if (hipHostGetDevicePointer((void**)&em->d_status_mirror, em->h_tagged, 0) != hipSuccess) em->d_status_mirror = nullptr;   // then: copies + events
    // launches of one pass: every length bucket, split into the sequences the grouped-column kernel
    // takes (no exception, or all of them within its virtual rows) and the rest
    const bool want_grouped = !sliced && prm->K <= 3u && c->use_grouped;
    // A shard of a sharded set plans its kernels as the whole set would: which rows a sequence is multiplied through
    // (mixed or uniform) decides the last bit of its responsibilities, so the choice follows the GLOBAL size the caller
    // names (n_seqs_bound / n_seqs_global; this shard's own count when it names neither) and nothing about the shard:
    // every length class of a set of `plan_n` sequences is planned as a launch of that many (a class that holds a small
    // part of a large set pays the larger tables' few microseconds per launch; an estimate of the class's global share
    // from this shard's own mix of lengths could differ between ranks next to the threshold).  The other input of the
    // plan, whether most sequences of a class carry exceptions, is the shard's own: a property of the data that holds
    // for every shard alike on double-stranded sets (each sequence has its strand junction) and on clean single-stranded ones.
    const uint64_t plan_n = std::max<uint64_t>(std::max<uint64_t>(prm->n_seqs_bound, prm->n_seqs_global), seqs->n);
    for (auto& b : seqs->buckets) {
        if (b.mclass == kLongClass || global_tables) {       // beyond the length classes / tables beyond LDS: long_seq.hip
            EmBucket eb;
            eb.mclass = kLongClass; eb.count = b.count; eb.d_idx = b.d_idx;
            eb.work = b.mclass == kLongClass ? b.work : (double)b.count * kMClasses[b.mclass];
            em->ebuckets.push_back(eb);
            prime(eb);
            continue;
        }
        const int Mcls = kMClasses[b.mclass];
        const uint32_t threads = default_threads(c, b.mclass);
        GrpGeom gg{};
        uint32_t glogc = 0, gG = 0;
        const ExcK::XRec* xr = nullptr;
        std::vector<uint32_t> yes, no;
        uint32_t glayout = 0;
        // do most sequences of this bucket carry exceptions (a double-stranded set: all of them)?
        std::atomic<size_t> with_exc_a{0};
        host_ranges(b.count, [&](uint64_t i0, uint64_t i1) {
            size_t cnt = 0;
            for (uint64_t i = i0; i < i1; i++) {
                const uint32_t n = b.d_idx ? b.h_idx[i] : (uint32_t)i;
                cnt += em->exc->h_off[n + 1] != em->exc->h_off[n];
            }
            with_exc_a.fetch_add(cnt, std::memory_order_relaxed);
        });
        const size_t with_exc = with_exc_a.load();
        if (want_grouped && grp_supported_class(Mcls, prm->K) &&
            grp_plan(prm->K, prm->W, Mcls, std::min(threads, grp_max_threads(Mcls)) / 64u, 2 * with_exc > b.count, plan_n * (uint64_t)Mcls >= 40000ull * 7ull, c->group_size, c->group_layout, &gG, &glogc, &glayout) &&
            grp_geometry(prm->K, prm->W, gG, Mcls, std::min(threads, grp_max_threads(Mcls)) / 64u, true, glogc, glayout, &gg)) {
            { EmBucket pb; pb.mclass = b.mclass; pb.grouped = true; pb.logc = glogc; pb.G = gG; pb.layout = glayout; prime(pb); }
            if ((rc = xrec_for_group(seqs, prm->K, gG, em->exc, &xr))) return fail(rc);
            // exceptions within the virtual rows for them, and clear of the rows for the LW1 edge
            auto capable = [&](uint32_t n) {
                const uint32_t B = xr->h_B[n];
                return B == 0u || (B <= gg.Bj && (gg.np != 0u || xr->h_lo[n] + B + gg.G <= seqs->h_len[n] - prm->W + 1u));
            };
            std::atomic<bool> all_capable{true};
            host_ranges(b.count, [&](uint64_t i0, uint64_t i1) {
                for (uint64_t i = i0; i < i1 && all_capable.load(std::memory_order_relaxed); i++)
                    if (!capable(b.d_idx ? b.h_idx[i] : (uint32_t)i)) all_capable.store(false, std::memory_order_relaxed);
            });
            const bool all = all_capable.load();
            if (!all)
                for (uint32_t i = 0; i < b.count; i++) {
                    const uint32_t n = b.d_idx ? b.h_idx[i] : i;
                    (capable(n) ? yes : no).push_back(n);
                }
            EmBucket eb;
            eb.mclass = b.mclass; eb.grouped = true; eb.logc = glogc; eb.G = gG; eb.layout = glayout; eb.d_xrec = xr->d_xrec;
            if (all) { eb.count = b.count; eb.d_idx = b.d_idx; }
            else {
                uint32_t* d = nullptr;
                if ((rc = dev_upload(c, &d, yes.data(), yes.size()))) return fail(rc);
                em->owned_idx.push_back(d);
                eb.count = (uint32_t)yes.size(); eb.d_idx = d;
            }
            eb.work = (double)eb.count * Mcls * 0.6;       // grouped passes cost about 60 % per sequence
            if (eb.count) em->ebuckets.push_back(eb);
            if (all) continue;
        }
        EmBucket eb;
        eb.mclass = b.mclass;
        if (!no.empty()) {
            uint32_t* d = nullptr;
            if ((rc = dev_upload(c, &d, no.data(), no.size()))) return fail(rc);
            em->owned_idx.push_back(d);
            eb.count = (uint32_t)no.size(); eb.d_idx = d;
        } else { eb.count = b.count; eb.d_idx = b.d_idx; }
        eb.work = (double)eb.count * Mcls;
        em->ebuckets.push_back(eb);
        prime(eb);
    }
    if (!em->owned_idx.empty() && hipStreamSynchronize(st) != hipSuccess) { set_error("stream sync failed"); return fail(BAMM_ERR_HIP); }
    // launch geometry: blocks split over the launches in proportion to their work
    double total_work = 0;
    for (auto& b : em->ebuckets) total_work += b.work;
    em->total_blocks = 0;
    for (auto& b : em->ebuckets) {
        if (b.mclass == kLongClass) {                        // a workgroup per sequence
            b.blocks = std::min(b.count, (uint32_t)std::max(1, c->num_cus) * 8u);
            em->total_blocks += b.blocks;
            continue;
        }
        const uint32_t threads = bucket_threads(c, b);
        // 16 waves per CU saturate the LDS pipe (tools/lds_bench2.hip); the LDS left over goes
        // into private copies of the count table
        const uint32_t blocks_per_cu = sliced ? 1u : std::max(1u, 1024u / threads);
        if (!b.grouped) {
            // sparse M-step scratch (per wave) competes with the private copies for LDS; it is only
            // enabled when at least 4 copies survive next to it
            const int Mcls = kMClasses[b.mclass];
            size_t scratch = sliced ? 0 : sparse_wave_bytes(Mcls) * (threads / 64u);
            uint32_t cap = sliced ? 0u : sparse_cap_for(Mcls);
            if (!c->use_sparse) { cap = 0; scratch = 0; }
            if (cap && (em_lds_bytes(prm->W, Y, true, 0, scratch) > kLds / blocks_per_cu ||
                        pick_log_copies(prm->W, Y, blocks_per_cu, scratch) + 1 < pick_log_copies(prm->W, Y, blocks_per_cu, 0))) {
                cap = 0;
                scratch = 0;
            }
            b.sparse_cap = cap;
            b.sparse_bytes = (uint32_t)(cap ? sparse_wave_bytes(Mcls) : 0);
            b.logc = sliced ? 0u : pick_log_copies(prm->W, Y, blocks_per_cu, scratch);
        }
        const uint32_t per_cu = b.grouped ? 1u : blocks_per_cu;
        const uint32_t all = c->blocks ? c->blocks : (uint32_t)std::max(1, c->num_cus) * per_cu;
        uint32_t nb = (uint32_t)std::max(1.0, std::floor(all * (b.work / total_work) + 0.5));
        const uint32_t waves_per_block = threads / 64u;
        nb = std::min(nb, (b.count + waves_per_block - 1) / waves_per_block);
        nb = std::max(nb, 1u);
        b.blocks = nb;
        em->total_blocks += nb;
    }
    // fused updates (update_kernel.h): inside iterate() / optimize() the model update of pass p runs in the block
    // prologue of pass p+1's FIRST launch -- a grouped kernel whose count tables leave room for the update's scratch
    if (!sliced && c->use_fused_update && update_fits_lds(prm->K, prm->W) && prm->K <= 2u) {
        size_t best = em->ebuckets.size();
        for (size_t i = 0; i < em->ebuckets.size(); i++)
            if (em->ebuckets[i].grouped && kMClasses[em->ebuckets[i].mclass] <= BAMM_FUSE_MAX_M &&   // the classes built with the fused prologue
                (best == em->ebuckets.size() || em->ebuckets[i].count > em->ebuckets[best].count)) best = i;
        if (best < em->ebuckets.size()) {
            std::swap(em->ebuckets[0], em->ebuckets[best]);
            const EmBucket& eb = em->ebuckets[0];
            GrpGeom g{};
            const uint32_t threads = bucket_threads(c, eb);
            if (grp_geometry(prm->K, prm->W, eb.G, kMClasses[eb.mclass], threads / 64u, true, eb.logc, eb.layout, &g)) {
                const uint32_t need = (uint32_t)update_lds_bytes(prm->K, prm->W);
                const uint32_t s1_bytes = (prm->W * (Y + 1u) * 4u + 15u) & ~15u;
                // mixed rows: behind the staged single-column table, inside the count tables; uniform rows: the count table
                const uint32_t off = (eb.layout & 8u) ? g.off_s1 + s1_bytes : g.off_ng;
                const uint32_t end = (eb.layout & 8u) ? g.off_wave : g.off_n1;
                if (off + need <= end) {
                    em->fusable = true;
                    em->fuse_upd_off = off;
                    if ((rc = dev_alloc(&em->d_s_block, (size_t)eb.blocks * prm->W * (Y + 1u)))) return fail(rc);
                }
            }
        }
    }
    if (sliced && em->e_fused && c->use_e_list) {
        if ((rc = scratch_alloc(c, &em->d_list_r, (size_t)seqs->total_len)) || (rc = scratch_alloc(c, &em->d_list_p, (size_t)seqs->total_len)) ||
            (rc = dev_alloc(&em->d_list_n, (size_t)seqs->n))) return fail(rc);
        if (hipMemsetAsync(em->d_list_n, 0, (seqs->n ? seqs->n : 1) * sizeof(uint32_t), st) != hipSuccess) { set_error("hipMemsetAsync failed"); return fail(BAMM_ERR_HIP); }
        // lists or dense r, per pass: the first pass of a handle takes the dense flavour (nothing is known yet: the
        // counter starts saturated), later ones lists once fewer than list_threshold_pct of the windows are non-zero
        if ((rc = dev_alloc(&em->d_nnz, 2))) return fail(rc);
        const unsigned long long start[2] = {~0ull, 0ull};
        if (hipMemcpyAsync(em->d_nnz, start, sizeof start, hipMemcpyHostToDevice, st) != hipSuccess) { set_error("hipMemcpyAsync failed"); return fail(BAMM_ERR_HIP); }
        unsigned long long windows = 0;
        for (uint64_t n = 0; n < seqs->n; n++) windows += seqs->h_len[n] - prm->W + 1u;
        if (seq_mask && seqs->n) windows = (unsigned long long)((double)windows * (double)em->n_active / (double)seqs->n);
        em->nnz_limit = windows / 100u * c->list_threshold_pct;
        em->adaptive_lists = c->use_adaptive_lists;
    }
    if ((rc = launch_make_s(em->d_v, em->d_vbg, prm->K, prm->W, em->Kbg, em->d_s, st))) return fail(rc);
    em->s_last = em->d_s;
    em->q_last = em->d_q;
    if (hipStreamSynchronize(st) != hipSuccess) { set_error("stream sync failed in bamm_em_create"); return fail(BAMM_ERR_HIP); }
    *out = em;
    return BAMM_OK;
}

int bamm_em_set_allreduce(bamm_em* em, bamm_allreduce_fn fn, void* user) {
    if (!em) { set_error("null em"); return BAMM_ERR_ARG; }
    // The int64 sums stay below 2^62 as long as (sequences summed over ALL ranks) x 2^fix_shift does; the unit was
    // chosen from this handle's own count unless the caller named a bound.  A callback says nothing about the world
    // behind it: up to 64 ranks of this size are assumed, beyond that the bound is required.
    if (fn && !em->prm.n_seqs_bound && !em->prm.n_seqs_global && em->seqs->n > (uint64_t(1) << 16)) {
        set_error("a shard of %llu sequences behind an all-reduce callback needs bamm_em_params.n_seqs_bound (all ranks together; "
                  "the unit of the integer accumulator must be the same on every rank and sized for their sum)", (unsigned long long)em->seqs->n);
        return BAMM_ERR_ARG;
    }
    em->allreduce = fn;
    em->allreduce_user = user;
    return BAMM_OK;
}

int bamm_em_set_comm(bamm_em* em, bamm_comm* comm) {
    if (!em) { set_error("null em"); return BAMM_ERR_ARG; }
    if (comm && comm_ctx(comm) != em->ctx) { set_error("the communicator belongs to another context"); return BAMM_ERR_ARG; }
    em->comm_verified = false;                                // checked with the peers in front of the first pass (verify_comm)
    em->comm = comm;
    return BAMM_OK;
}

int bamm_em_estep(bamm_em* em) {
    if (!em) { set_error("null em"); return BAMM_ERR_ARG; }
    if (int vrc = verify_comm(em)) return vrc;
    // s already reflects the current v (made at create / by the last update): E only
    int rc = run_accumulate(em, false);
    if (rc) return rc;
    if ((rc = run_allreduce(em))) return rc;
    if ((rc = launch_stat_only(em->d_acc, (uint32_t)em->cells, em->d_status, em->ctx->stream))) return rc;
    em->acc_dirty = false;
    em->estep_done = true;
    return BAMM_OK;
}

int bamm_em_mstep(bamm_em* em) {
    if (!em) { set_error("null em"); return BAMM_ERR_ARG; }
    if (int vrc = verify_comm(em)) return vrc;
    if (!em->estep_done) { set_error("MStep needs the responsibilities of a preceding EStep"); return BAMM_ERR_STATE; }
    // the responsibilities are a pure function of (s, q), both unchanged since the EStep:
    // recompute them on the fly while accumulating counts instead of storing N*L floats
    const int32_t oq = em->prm.optimize_q;
    em->prm.optimize_q = 0;                           // EM::MStep never touches q
    int rc = run_accumulate(em, true, true);          // with the (s, q) the EStep saw, even if q moved since
    if (!rc) rc = run_allreduce(em);
    if (!rc) rc = run_update(em, false);
    em->prm.optimize_q = oq;
    return rc;
}

int bamm_em_optimize_q(bamm_em* em) {
    if (!em) { set_error("null em"); return BAMM_ERR_ARG; }
    int rc = fetch_status(em);
    if (rc) return rc;
    const double nseq = em->prm.n_seqs_global ? (double)em->prm.n_seqs_global : (double)em->h_status[5];
    const float q = (float)((nseq - (double)em->h_status[4] + 1.0) / (nseq + 2.0));   // EM.cpp:515
    // the slot the last EStep read stays intact for MStep()/getR(), in whatever order the caller
    // runs MStep() and optimize_q() (EM.cpp:93-99 has MStep first)
    float* slot = q_write_slot(em);
    BAMM_HIP(hipSetDevice(em->ctx->device));
    BAMM_HIP(hipMemcpyAsync(slot, &q, sizeof(float), hipMemcpyHostToDevice, em->ctx->stream));
    BAMM_HIP(hipStreamSynchronize(em->ctx->stream));
    em->d_q = slot;
    return BAMM_OK;
}

int bamm_em_accumulate(bamm_em* em) {
    if (!em) { set_error("null em"); return BAMM_ERR_ARG; }
    return run_accumulate(em, true);
}

int bamm_em_reduce_buffer(bamm_em* em, void** dev_ptr, uint64_t* n_words) {
    if (!em || !dev_ptr || !n_words) { set_error("bad argument"); return BAMM_ERR_ARG; }
    *dev_ptr = em->d_acc;
    *n_words = em->cells + 3;
    return BAMM_OK;
}

int bamm_em_set_reduce_buffer(bamm_em* em, void* dev_ptr, uint64_t n_words) {
    if (!em || !dev_ptr) { set_error("bad argument"); return BAMM_ERR_ARG; }
    if (n_words < em->cells + 3) {
        set_error("reduce buffer holds %llu 64-bit words, %llu needed", (unsigned long long)n_words, (unsigned long long)(em->cells + 3));
        return BAMM_ERR_ARG;
    }
    BAMM_HIP(hipSetDevice(em->ctx->device));
    BAMM_HIP(hipStreamSynchronize(em->ctx->stream));
    (void)hipFree(em->d_acc_ring);                           // a caller-owned accumulator is one slot: no fused updates
    em->d_acc_ring = nullptr; em->acc_cur = 0; em->ring_prev_dirty = false; em->fusable = false;
    em->d_acc = static_cast<long long*>(dev_ptr);
    em->acc_external = true;
    em->acc_dirty = false;
    BAMM_HIP(hipMemsetAsync(em->d_acc, 0, (em->cells + 3) * sizeof(long long), em->ctx->stream));
    return BAMM_OK;
}

int bamm_em_update(bamm_em* em) {
    if (!em) { set_error("null em"); return BAMM_ERR_ARG; }
    // hand-driven passes have no optimize() call to be local to: the handle's first five updates
    return run_update(em, em->host_iteration < 5u);
}

int bamm_em_iterate(bamm_em* em, uint32_t n) {
    if (!em) { set_error("null em"); return BAMM_ERR_ARG; }
    if (int vrc = verify_comm(em)) return vrc;
    em->events_used = 0; em->pass_no = 0; em->region_open = false;
    TimedRegionCloser closer{em};                            // (an error return)
    // fusable handles: the update of pass i runs in the prologue of pass i+1's first kernel (one launch and one
    // collective per iteration); the last pass's update is a k_update launch, so the handle is in the same state
    // at every API boundary whichever way its updates ran
    for (uint32_t i = 0; i < n; i++) {
        int rc = run_accumulate(em, true, false, false, (em->fusable && i > 0u) ? (int)(i - 1u < 5u) : -1);
        if (!rc && i + 1u == n) rc = close_timed_region(em); // a whole-call interval ends behind the last pass's sequence kernel
        if (!rc) rc = run_allreduce(em);
        if (!rc && (!em->fusable || i + 1u == n)) rc = run_update(em, i < 5u);
        if (rc) { em->acc_dirty = true; return rc; }
    }
    return BAMM_OK;
}

int bamm_em_optimize(bamm_em* em, uint32_t* iterations) {
    if (!em) { set_error("null em"); return BAMM_ERR_ARG; }
    if (int vrc = verify_comm(em)) return vrc;
    em->events_used = 0; em->pass_no = 0; em->region_open = false;
    TimedRegionCloser closer{em};
    if (iterations) *iterations = 0;
    const uint32_t max_it = em->prm.max_iterations;
    if (max_it == 0) return BAMM_OK;
    int rc = use_device(em->ctx);
    if (rc) return rc;
    hipStream_t st = em->ctx->stream;
    // The stop rule (EM.cpp:117-118) needs (llh, v_diff) of a pass on the host: a read-back and a stream
    // synchronisation per pass, 13 us during which the GPU idles (half of a pass at 300 sequences, a sixth at 50k).
    // So the loop runs AHEAD of the numbers it waits for.  The update evaluates the same rule on the device and
    // raises a flag; whatever was enqueued behind a raised flag does nothing (every kernel returns at entry), so the
    // model is exactly what the stopping pass left, and the host takes back the bookkeeping of the work that did
    // not happen (prepare_update keeps a snapshot per update).
    //
    // The stream carries UNITS.  Plain handles: unit u = pass u + all-reduce + k_update(u); status(u) is there when
    // unit u is.  Fusable handles (update_kernel.h): unit u = [update(u-1) in the prologue of] pass u + all-reduce,
    // and one last unit max_it + 1 = k_update(max_it); status(u) is there when unit u + 1 is (lag 1).  A fused
    // update that fires the rule ends every block of its kernel before the pass: same model, same trace.
    if (!em->d_stop && (rc = dev_alloc(&em->d_stop, 1))) return rc;
    for (hipEvent_t& e : em->opt_events)
        if (!e) BAMM_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    BAMM_HIP(hipMemsetAsync(em->d_stop, 0, sizeof(uint32_t), st));
    const uint32_t lag = em->fusable ? 1u : 0u;
    const uint32_t units = max_it + lag;
    // The status of update i reaches the host through pinned memory the update's writer stores into (UpdateArgs::
    // status_mirror) as six self-validating words tagged with i: the host POLLS them.  An event per unit, which this loop
    // used to record and wait for, costs the stream 4 us per pass (optimize() against iterate(): +4.5 us at every size up to
    // 50k sequences, profiles/r05_optimize_vs_iterate.txt); events remain the fallback where the mirror could not be mapped.
    const bool poll = em->d_status_mirror != nullptr;
    if (poll) memset(em->h_tagged, 0, 16 * sizeof(unsigned long long));         // no tag of an earlier call (tags start at 1)
    const uint32_t first_update = em->host_iteration;       // update(i) of this call is the handle's update first_update + i
    em->stop_arg = em->d_stop;
    em->opt_llh_prev = em->llh_prev;
    auto enqueue_unit = [&](uint32_t u) -> int {            // 1-based
        int r = BAMM_OK;
        if (u <= max_it) {
            if (lag && u >= 2u) {
                em->opt_iteration = u - 1u;
                r = run_accumulate(em, true, false, false, (int)(u - 1u <= 5u));     // EM.cpp:99 for update(u-1)
            } else {
                r = run_accumulate(em, true);
            }
            if (!r) r = run_allreduce(em);
            if (!r && !lag) { em->opt_iteration = u; r = run_update(em, u <= 5u); }    // EM.cpp:99
        } else {
            em->opt_iteration = max_it;
            r = run_update(em, max_it <= 5u);
        }
        if (r) return r;
        const uint32_t upd = u - lag;                        // the update this unit carried (0: none)
        if (!poll) {
            if (upd >= 1u) BAMM_HIP(hipMemcpyAsync(em->h_status + 8 + 8 * (upd & 1u), em->d_status, 8 * sizeof(float), hipMemcpyDeviceToHost, st));
            BAMM_HIP(hipEventRecord(em->opt_events[u & 1u], st));
        }
        return BAMM_OK;
    };
    // every exit: the kernels stop looking at the flag; sums nobody consumed are cleared before the next pass
    auto leave = [&](int r) { em->stop_arg = nullptr; if (r) em->acc_dirty = true; return r; };
    uint32_t enqueued = 0, done = 0;
    float llh = em->llh_prev;
    for (;;) {                                                              // EM.cpp:81
        done++;
        while (enqueued < std::min(units, done + lag + 1u)) {               // the unit with status(done) and one beyond it
            if ((rc = enqueue_unit(enqueued + 1u))) return leave(rc);
            enqueued++;
        }
        const float* hs = em->h_status + 8 + 8 * (done & 1u);
        float polled[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (poll) {
            const volatile unsigned long long* slot = em->h_tagged + 8 * (done & 1u);
            auto arrived = [&] {
                for (int i = 0; i < 6; i++) {
                    const unsigned long long w = slot[i];
                    if ((uint32_t)(w >> 32) != done) return false;
                    const uint32_t bits = (uint32_t)w;
                    memcpy(&polled[i], &bits, sizeof(float));
                }
                return true;
            };
            hs = polled;
            for (uint32_t spins = 1;; spins++) {
                if (arrived()) break;
                if ((spins & 2047u) == 0u) {                                    // now and then: is anything still running?
                    const hipError_t qs = hipStreamQuery(st);
                    if (qs == hipSuccess) {                                     // the stream is idle: the tag is there, or never will be
                        if (arrived()) break;
                        if (int cs = comm_still_sound(em)) return leave(cs);   // (a block gave up waiting for a peer: every later launch did nothing)
                        set_error("optimize(): pass %u ended without reporting its status (a kernel of the pass failed?)", done);
                        return leave(BAMM_ERR_HIP);
                    }
                    if (qs != hipErrorNotReady) { (void)hipGetLastError(); set_error("optimize(): %s", hipGetErrorString(qs)); return leave(BAMM_ERR_HIP); }
                    if (em->comm && comm_aborted(em->comm)) { set_error("the communicator was aborted while optimize() was waiting for pass %u", done); return leave(BAMM_ERR_COMM); }
                }
                __builtin_ia32_pause();
            }
            std::atomic_thread_fence(std::memory_order_acquire);
        } else if (hipEventSynchronize(em->opt_events[(done + lag) & 1u]) != hipSuccess) {
            set_error("hipEventSynchronize failed in optimize()");
            return leave(BAMM_ERR_HIP);
        }
        const float llh_prev = llh;
        llh = hs[0];
        const float v_diff = hs[1];
        bool iterate = true;
        if (v_diff < em->prm.epsilon) iterate = false;                      // EM.cpp:117
        if (llh - llh_prev < 0 && done > 10) iterate = false;               // EM.cpp:118
        if (!iterate || done == max_it) {
            memcpy(em->h_status, hs, 8 * sizeof(float));
            if (!iterate) {
                // whatever was enqueued behind update(done) found the flag raised and did nothing: back to the snapshot
                // prepare_update took right after update(done) (buffers, iteration count, kernel-timing samples)
                restore_book(em, em->books[(first_update + done) & 3u]);
                if (lag) em->acc_dirty = true;                              // the ring slot the fused update read was never cleared
            }
            break;
        }
    }
    em->stop_arg = nullptr;
    em->llh_prev = llh;
    if (iterations) *iterations = done;
    return BAMM_OK;
}

int bamm_em_mask(bamm_em* em, float f, uint32_t* iterations, float* cutoff, uint64_t* listed) {
    if (!em) { set_error("null em"); return BAMM_ERR_ARG; }
    if (int vrc = verify_comm(em)) return vrc;
    bamm_seqs* s = em->seqs;
    em->pass_summed_in_kernel = false;                       // EM::mask's kernels carry no tail: every one of its sums goes through the communicator
    if (!(f > 0.0f && f < 1.0f)) { set_error("bamm_em_mask: fraction %g outside (0,1)", (double)f); return BAMM_ERR_ARG; }
    if (em->n_active == 0) { set_error("bamm_em_mask: no sequences (the reference indexes an empty array, EM.cpp:343)"); return BAMM_ERR_ARG; }
    if (em->prm.W < 2) { set_error("bamm_em_mask: W=1 reads past pos_[n] in the reference (EM.cpp:416)"); return BAMM_ERR_UNSUPPORTED; }
    if (em->prm.optimize_q && (em->allreduce || em->comm)) {
        set_error("bamm_em_mask: optimize_q re-estimates q after every sequence (EM.cpp:321), a serial chain that cannot be sharded");
        return BAMM_ERR_UNSUPPORTED;
    }
    if (em->host_iteration != 0 || em->estep_done) {
        set_error("bamm_em_mask: the handle has already run E/M passes; the reference calls mask() on a fresh EM only");
        return BAMM_ERR_STATE;
    }
    if (int rcc = use_device(em->ctx)) return rcc;
    if (int rcc = clean_accumulator(em)) return rcc;          // sums nobody consumed (bamm_em_accumulate without an update, a getR replay)
    const size_t kLds = 160 * 1024;
    const uint32_t W = em->prm.W, Y = em->Y;
    auto round16 = [](size_t x) { return (x + 15) & ~(size_t)15; };
    const size_t s_bytes = round16((size_t)W * (Y + 1) * sizeof(float));
    // M-step: as many columns per launch as fit next to one wave's arrays.  Sequences whose arrays (10 bytes per
    // position) do not fit beside one column's counts keep them in a global scratch region per wave instead
    // (~16 000 positions at k = 2; the window lists are 32 bits wide there, so any length goes): slower, same
    // arithmetic in the same order.  Orders whose count column alone exceeds the LDS (k >= 7) add the listed windows
    // straight into the pass's accumulator.  The reference has neither limit (EM.cpp:261-503).
    const bool direct = round16((size_t)Y * 8) > kLds;
    const bool wave_global = direct || mask_wave_bytes(s->max_len, false) + round16((size_t)Y * 8) > kLds || s->max_len > 65535u;
    const size_t wave_bytes = mask_wave_bytes(s->max_len, wave_global);
    if (wave_bytes > 0xffffffffull) { set_error("bamm_em_mask: sequences beyond 2^28 positions"); return BAMM_ERR_UNSUPPORTED; }
    const bool s_in_lds = s_bytes <= 64 * 1024 && (wave_global || s_bytes + wave_bytes <= kLds);
    const size_t e_table = s_in_lds ? s_bytes : 0;
    uint32_t m_cols = direct ? W : (uint32_t)std::min<size_t>(W, (wave_global ? kLds / 2 : std::min(kLds / 2, kLds - wave_bytes)) / ((size_t)Y * 8));
    m_cols = std::max(1u, m_cols);
    const size_t m_table = direct ? 0 : round16((size_t)m_cols * Y * 8);
    auto waves_for = [&](size_t table) {
        return wave_global ? 4u : (uint32_t)std::max<size_t>(1, std::min<size_t>(4, (kLds - table) / wave_bytes));
    };
    const uint32_t init_table = (uint32_t)round16((size_t)W * 4 * sizeof(float));
    int rc = use_device(em->ctx);
    if (rc) return rc;
    hipStream_t st = em->ctx->stream;
    const uint32_t e_waves = waves_for(e_table), m_waves = waves_for(m_table);
    // 16 waves per CU (as the fused kernel), but no more partial tables than 64 MiB worth
    // (arrays in global memory: at most 2048 waves' worth of them)
    // ... and no more than 8 GiB of them: a launch has at most cus * 8 blocks of 4 waves
    const uint32_t cus = wave_global ? (uint32_t)std::max<size_t>(1, std::min<size_t>(std::min(64u, (uint32_t)std::max(1, em->ctx->num_cus)), ((size_t)8 << 30) / (32 * wave_bytes)))
                                     : (uint32_t)std::max(1, em->ctx->num_cus);
    const uint32_t per_cu = std::max(1u, 16u / std::min(e_waves, m_waves));
    const uint32_t cap_blocks = direct ? cus * 8u            // no partial tables at all
                                       : (uint32_t)std::max<size_t>(cus, std::min<size_t>((size_t)cus * per_cu, ((size_t)64 << 20) / (em->cells * 8)));
    const uint32_t mblocks = std::max(1u, std::min(((uint32_t)s->n + std::min(e_waves, m_waves) - 1) / std::min(e_waves, m_waves), cap_blocks));
    if (!em->d_mask_r) {
        em->mask_blocks = mblocks;
        if ((!direct && (rc = dev_alloc(&em->d_mask_partial_n, (size_t)mblocks * em->cells))) ||
            (rc = dev_alloc(&em->d_mask_partial_stat, (size_t)mblocks * 4)) ||
            (rc = scratch_alloc(em->ctx, &em->d_mask_r, (size_t)s->total_len)) ||
            (rc = dev_alloc(&em->d_mask_bits, (size_t)s->total_len / 32 + 2)) ||
            (rc = dev_alloc(&em->d_mask_hist, 2049)) || (rc = dev_alloc(&em->d_mask_sel, 1)) ||
            (em->prm.optimize_q && (rc = dev_alloc(&em->d_mask_qseq, (size_t)s->n))))
            return rc;
    }
    BAMM_HIP(hipMemsetAsync(em->d_mask_r, 0, (size_t)s->total_len * sizeof(float), st));
    BAMM_HIP(hipMemsetAsync(em->d_mask_bits, 0, ((size_t)s->total_len / 32 + 2) * sizeof(uint32_t), st));
    BAMM_HIP(hipMemsetAsync(em->d_mask_hist, 0, 2049 * sizeof(long long), st));
    BAMM_HIP(hipMemsetAsync(em->d_mask_sel, 0, sizeof(MaskSelect), st));

    Bucket all;                                              // every sequence in natural order
    all.count = (uint32_t)s->n;
    MaskKernelArgs a{};
    a.sv = make_view(s, em->exc, all, em->d_mask);
    a.K = em->prm.K; a.W = W; a.Y = Y;
    a.max_len = s->max_len;
    a.wave_bytes = (uint32_t)wave_bytes;
    a.v0 = em->d_v; a.vbg0 = em->d_vbg;
    a.q = em->d_q;
    a.q_seq = em->prm.optimize_q ? em->d_mask_qseq : nullptr;
    a.n_total = (float)em->n_active;
    a.fix_scale = ldexpf(1.0f, (int)em->fix_shift - 40);
    a.r = em->d_mask_r; a.bits = em->d_mask_bits; a.hist = em->d_mask_hist; a.sel = em->d_mask_sel;
    a.partial_n = em->d_mask_partial_n; a.partial_stat = em->d_mask_partial_stat;
    unsigned char* d_wave = nullptr;
    if (wave_global) {                                       // every launch below has at most cus * 8 blocks of 4 waves
        if ((rc = scratch_alloc(em->ctx, &d_wave, (size_t)std::max(cus * 8u, mblocks) * 4u * wave_bytes))) return rc;
        a.wave_scratch = d_wave;
    }
    struct WaveScratch { bamm_ctx* c; unsigned char* p; ~WaveScratch() { scratch_free(c, p); } } wave_guard{em->ctx, d_wave};

    auto blocks_for = [&](uint32_t waves, uint32_t cap) {
        const uint32_t need = ((uint32_t)s->n + waves - 1) / waves;
        return std::max(1u, std::min(need, cap));
    };
    // order-0 pass (EM.cpp:266-323)
    {
        a.table_bytes = init_table;
        const uint32_t waves = waves_for(init_table);
        if ((rc = launch_mask_init(a, em->prm.optimize_q != 0, blocks_for(waves, cus * 8), waves * 64, st))) return rc;
    }
    // cut-off (EM.cpp:329-343)
    for (int pass = 0; pass < 3; pass++) {
        if ((rc = launch_mask_hist(a, pass, blocks_for(4, cus * 8), st))) return rc;
        if ((rc = allreduce_words(em, em->d_mask_hist, 2049))) return rc;
        if ((rc = launch_mask_pick(a, pass, f, st))) return rc;
    }
    if ((rc = launch_mask_bits(a, blocks_for(4, cus * 8), st))) return rc;     // EM.cpp:345-356

    // EM over the listed windows (EM.cpp:373-494)
    const int32_t oq = em->prm.optimize_q;
    em->prm.optimize_q = 0;                                  // q is not touched inside this loop
    em->events_used = 0; em->pass_no = 0; em->region_open = false;
    TimedRegionCloser closer{em};
    bool iterate = true;
    uint32_t iteration = 0;
    float llh = em->llh_prev;
    while (iterate && iteration < em->prm.max_iterations && !rc) {
        iteration++;
        const float llh_prev = llh;
        a.s = em->d_s;
        a.q = em->d_q;
        if ((rc = record_event(em, true))) break;
        a.table_bytes = (uint32_t)e_table;
        rc = launch_mask_e(a, s_in_lds, mblocks, e_waves * 64, st);
        a.table_bytes = (uint32_t)m_table;
        for (uint32_t j0 = 0; j0 < W && !rc; j0 += m_cols) {
            a.j0 = j0; a.j1 = std::min(W, j0 + m_cols);
            a.acc_direct = direct ? em->d_acc : nullptr;
            rc = launch_mask_m(a, mblocks, m_waves * 64, st);
        }
        if (!rc) rc = record_event(em, false);
        if (!rc) rc = launch_reduce_partials(direct ? nullptr : em->d_mask_partial_n, em->d_mask_partial_stat, mblocks, W, Y, em->d_acc, st);
        if (!rc) rc = run_allreduce(em);
        if (!rc) rc = run_update(em, false);
        if (!rc) rc = fetch_status(em);
        if (rc) break;
        llh = em->h_status[0];
        if (em->h_status[1] < em->prm.epsilon) iterate = false;            // EM.cpp:488
        if (llh - llh_prev < 0 && iteration > 10) iterate = false;         // EM.cpp:489
    }
    em->prm.optimize_q = oq;
    if (rc) return rc;
    em->llh_prev = llh;
    em->mask_done = true;
    MaskSelect sel;
    BAMM_HIP(hipMemcpyAsync(&sel, em->d_mask_sel, sizeof(sel), hipMemcpyDeviceToHost, st));
    BAMM_HIP(hipStreamSynchronize(st));
    if (iterations) *iterations = iteration;
    if (cutoff) *cutoff = sel.cutoff;
    if (listed) *listed = sel.listed;
    return BAMM_OK;
}

static int copy_out(bamm_em* em, float* dst, const float* src, size_t count) {
    if (!em || !dst) { set_error("bad argument"); return BAMM_ERR_ARG; }
    BAMM_HIP(hipSetDevice(em->ctx->device));
    if (int rc = ctx_download(em->ctx, dst, src, count * sizeof(float))) return rc;
    BAMM_HIP(hipStreamSynchronize(em->ctx->stream));
    return comm_still_sound(em);
}

int bamm_em_get_v(bamm_em* em, float* v) { return copy_out(em, v, em ? em->d_v : nullptr, em ? em->vsz : 0); }
int bamm_em_get_counts(bamm_em* em, float* n) { return copy_out(em, n, em ? em->d_n : nullptr, em ? em->vsz : 0); }

int bamm_em_get_s(bamm_em* em, float* s) {
    if (!em || !s) { set_error("bad argument"); return BAMM_ERR_ARG; }
    const size_t Ys = em->Y + 1;
    std::vector<float> tmp((size_t)em->prm.W * Ys);
    int rc = copy_out(em, tmp.data(), em->d_s, tmp.size());
    if (rc) return rc;
    for (uint32_t y = 0; y < em->Y; y++)
        for (uint32_t j = 0; j < em->prm.W; j++) s[(size_t)y * em->prm.W + j] = tmp[(size_t)j * Ys + y];
    return BAMM_OK;
}

int bamm_em_get_q(bamm_em* em, float* q) { return copy_out(em, q, em ? em->d_q : nullptr, 1); }

int bamm_em_get_llh(bamm_em* em, float* llh) {
    if (!em || !llh) { set_error("bad argument"); return BAMM_ERR_ARG; }
    int rc = fetch_status(em);
    if (rc) return rc;
    *llh = em->h_status[0];
    return BAMM_OK;
}

int bamm_em_get_vdiff(bamm_em* em, float* vd) {
    if (!em || !vd) { set_error("bad argument"); return BAMM_ERR_ARG; }
    int rc = fetch_status(em);
    if (rc) return rc;
    *vd = em->h_status[1];
    return BAMM_OK;
}

int bamm_em_get_iteration(bamm_em* em, uint32_t* it) {
    if (!em || !it) { set_error("bad argument"); return BAMM_ERR_ARG; }
    *it = em->host_iteration;
    return BAMM_OK;
}

int bamm_em_get_r(bamm_em* em, uint64_t begin, uint64_t end, float* out, uint64_t out_cap) {
    if (!em || !out || begin > end || end > em->seqs->n) { set_error("bamm_em_get_r: bad range"); return BAMM_ERR_ARG; }
    bamm_seqs* s = em->seqs;
    const uint64_t base = s->h_pos_off[begin], total = s->h_pos_off[end] - base;
    if (out_cap < total) { set_error("bamm_em_get_r: output holds %llu floats, %llu needed", (unsigned long long)out_cap, (unsigned long long)total); return BAMM_ERR_ARG; }
    if (total == 0) return BAMM_OK;
    BAMM_HIP(hipSetDevice(em->ctx->device));
    hipStream_t st = em->ctx->stream;
    if (em->mask_done) {                                    // EM::mask keeps r_ materialised (EM.cpp:409-430)
        if (int rc = ctx_download(em->ctx, out, em->d_mask_r + base, total * sizeof(float))) return rc;
        BAMM_HIP(hipStreamSynchronize(st));
        return BAMM_OK;
    }
    if (em->sliced) {
        // the sliced E pass leaves r per position slot p (window start i = p-W+1) in d_state;
        // the reference's index is L-W-i = L-1-p (EM.cpp:173)
        uint8_t* saved_mask = em->d_mask;
        em->d_mask = nullptr;                              // masked-out sequences still have an r
        const uint32_t used = em->events_used, pass_no = em->pass_no;
        int rc2 = run_accumulate(em, false, true, true);       // dense r, in the reference's layout
        em->events_used = used; em->pass_no = pass_no;
        em->d_mask = saved_mask;
        if (rc2) return rc2;
        if (int rc3 = ctx_download(em->ctx, out, em->d_state + base, total * sizeof(float))) return rc3;
        BAMM_HIP(hipStreamSynchronize(st));
        if (!em->e_fused)
            for (uint64_t n = begin; n < end; n++) {
                float* r = out + (s->h_pos_off[n] - base);
                std::reverse(r, r + s->h_len[n]);
            }
        return BAMM_OK;
    }
    float* d_r = nullptr;
    int rc = scratch_alloc(em->ctx, &d_r, total);
    if (rc) return rc;
    if (hipMemsetAsync(d_r, 0, total * sizeof(float), st) != hipSuccess) { scratch_free(em->ctx, d_r); set_error("hipMemsetAsync failed"); return BAMM_ERR_HIP; }
    for (size_t b = 0; b < em->ebuckets.size() && !rc; b++) {
        const EmBucket& bk = em->ebuckets[b];
        EmKernelArgs a{};
        a.sv = make_view(s, em->exc, bk, nullptr);        // masked-out sequences still have an r in the reference
        a.K = em->prm.K; a.W = em->prm.W; a.Y = em->Y;
        a.s = em->s_last; a.q = em->q_last;                 // the E pass the caller last ran (EM.cpp:521)
        a.acc = nullptr;                                    // responsibilities only
        a.r_out = d_r; a.r_base = base; a.seq_begin = (uint32_t)begin; a.seq_end = (uint32_t)end;
        a.fix_scale = 1.0f;
        if (bk.mclass == kLongClass) { rc = launch_long_em(a, false, true, false, bk.blocks, st); continue; }
        rc = launch_fused(em, bk, false, true, a, bucket_threads(em->ctx, bk), st);
    }
    if (!rc) {
        rc = ctx_download(em->ctx, out, d_r, total * sizeof(float));
        if (!rc && hipStreamSynchronize(st) != hipSuccess) { set_error("copy of r failed"); rc = BAMM_ERR_HIP; }
    }
    scratch_free(em->ctx, d_r);
    return rc;
}

int bamm_em_get_trace(bamm_em* em, float* llh, float* v_diff, float* q, uint32_t cap, uint32_t* n) {
    if (!em || !n) { set_error("bad argument"); return BAMM_ERR_ARG; }
    const uint32_t avail = std::min(em->host_iteration, em->prm.max_iterations);
    *n = avail;
    const uint32_t m = std::min(avail, cap);
    if (m == 0) return BAMM_OK;
    std::vector<float> tmp((size_t)m * 3);
    int rc = copy_out(em, tmp.data(), em->d_trace, tmp.size());
    if (rc) return rc;
    for (uint32_t i = 0; i < m; i++) {
        if (llh) llh[i] = tmp[(size_t)i * 3 + 0];
        if (v_diff) v_diff[i] = tmp[(size_t)i * 3 + 1];
        if (q) q[i] = tmp[(size_t)i * 3 + 2];
    }
    return BAMM_OK;
}

int bamm_em_plan(bamm_em* em, uint64_t* grouped_seqs, uint64_t* percolumn_seqs, uint32_t* launches) {
    if (!em) { set_error("null em"); return BAMM_ERR_ARG; }
    uint64_t g = 0, o = 0;
    for (auto& b : em->ebuckets) (b.grouped ? g : o) += b.count;
    if (grouped_seqs) *grouped_seqs = g;
    if (percolumn_seqs) *percolumn_seqs = o;
    if (launches) *launches = (uint32_t)em->ebuckets.size();
    return BAMM_OK;
}

int bamm_em_plan_mixed(bamm_em* em, uint64_t* mixed_seqs) {
    if (!em || !mixed_seqs) { set_error("bad argument"); return BAMM_ERR_ARG; }
    uint64_t m = 0;
    for (auto& b : em->ebuckets) if (b.grouped && (b.layout & 8u)) m += b.count;
    *mixed_seqs = m;
    return BAMM_OK;
}

int bamm_em_comm_mode(bamm_em* em, int* mode, char* note, size_t note_cap) {
    if (!em || !mode) { set_error("bad argument"); return BAMM_ERR_ARG; }
    if (int vrc = verify_comm(em)) return vrc;                // collective on first use, like the first pass would be
    *mode = em->peer_on ? 2 : ((em->comm || em->allreduce) ? 1 : 0);
    if (note && note_cap) snprintf(note, note_cap, "%s", em->peer_note.c_str());
    return BAMM_OK;
}

int bamm_em_set_kernel_timing(bamm_em* em, uint32_t every) {
    if (!em) { set_error("null EM handle"); return BAMM_ERR_ARG; }
    em->timing_every = every;
    return BAMM_OK;
}

int bamm_em_kernel_time(bamm_em* em, float* total_ms, uint32_t* launches) {
    if (!em || !total_ms || !launches) { set_error("bad argument"); return BAMM_ERR_ARG; }
    if (int rc = close_timed_region(em)) return rc;           // hand-driven passes (bamm_em_accumulate) in whole-call mode
    BAMM_HIP(hipStreamSynchronize(em->ctx->stream));
    float acc = 0.0f;
    uint32_t passes = 0;
    for (uint32_t i = 0; i < em->events_used; i++) {
        float ms = 0.0f;
        BAMM_HIP(hipEventElapsedTime(&ms, em->events[i].first, em->events[i].second));
        acc += ms;
        passes += em->event_passes[i];
    }
    *total_ms = acc;
    *launches = passes;
    return BAMM_OK;
}

// ------------------------------------------------------------------------------ scorer -----
int bamm_seed_from_pwm(bamm_ctx* c, bamm_seqs* s, uint32_t K, uint32_t W, const float* score, float q, const double* u,
                       int32_t* counts, uint32_t* z) {
    if (!c || !s || !score || !u || !counts) { set_error("bamm_seed_from_pwm: null argument"); return BAMM_ERR_ARG; }
    if (K > BAMM_MAX_ORDER || W == 0) { set_error("bamm_seed_from_pwm: bad K/W"); return BAMM_ERR_ARG; }
    if (s->ctx != c) { set_error("sequence set belongs to another context"); return BAMM_ERR_ARG; }
    const size_t vsz = v_size(K, W);
    std::fill(counts, counts + vsz, 0);
    if (s->n == 0) return BAMM_OK;
    BAMM_HIP(hipSetDevice(c->device));
    hipStream_t st = c->stream;
    ExcK* exc = nullptr;
    int rc = exceptions_for_order(s, K, &exc);
    if (rc) return rc;
    float* d_score = nullptr;
    double* d_u = nullptr;
    int* d_counts = nullptr;
    uint32_t* d_z = nullptr;
    unsigned char* d_wave = nullptr;
    auto cleanup = [&]() { (void)hipFree(d_score); (void)hipFree(d_u); (void)hipFree(d_counts); (void)hipFree(d_z); scratch_free(c, d_wave); };
    if ((rc = dev_upload(c, &d_score, score, (size_t)4 * W)) || (rc = dev_upload(c, &d_u, u, s->n)) ||
        (rc = dev_alloc(&d_counts, vsz)) || (z && (rc = dev_alloc(&d_z, s->n)))) { cleanup(); return rc; }
    if (hipMemsetAsync(d_counts, 0, vsz * sizeof(int), st) != hipSuccess) { set_error("hipMemsetAsync failed"); cleanup(); return BAMM_ERR_HIP; }
    SeedKernelArgs a{};
    Bucket all;                                              // every sequence, natural order
    all.count = (uint32_t)s->n;
    a.sv = make_view(s, exc, all, nullptr);
    a.K = K; a.W = W; a.Y = (uint32_t)ipow4(K + 1);
    a.max_len = s->max_len; a.vsize = (uint32_t)vsz;
    a.score = d_score; a.q = q; a.u = d_u; a.counts = d_counts; a.z_out = d_z;
    if (const size_t need = seed_global_scratch_bytes(a, (uint32_t)std::max(1, c->num_cus))) {   // sequences beyond the LDS plan
        if ((rc = scratch_alloc(c, &d_wave, need))) { cleanup(); return rc; }
        a.wave_scratch = d_wave;
    }
    rc = launch_seed_pwm(a, (uint32_t)std::max(1, c->num_cus), st);
    if (!rc) {
        hipError_t e = ctx_download(c, counts, d_counts, vsz * sizeof(int)) ? hipErrorUnknown : hipSuccess;
        if (e == hipSuccess && z && ctx_download(c, z, d_z, s->n * sizeof(uint32_t))) e = hipErrorUnknown;
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) { set_error("bamm_seed_from_pwm: copy failed: %s", hipGetErrorString(e)); rc = BAMM_ERR_HIP; }
    }
    cleanup();
    return rc;
}

int bamm_logodds(bamm_ctx* c, bamm_seqs* s, uint32_t K, uint32_t W, uint32_t bg_order, const float* v, const float* vbg,
                 float* mops, uint64_t mops_cap, float* zoops, uint64_t* z) {
    return bamm_logodds_subset(c, s, nullptr, K, W, bg_order, v, vbg, mops, mops_cap, zoops, z);
}

int bamm_logodds_subset(bamm_ctx* c, bamm_seqs* s, const uint8_t* seq_mask, uint32_t K, uint32_t W, uint32_t bg_order,
                        const float* v, const float* vbg, float* mops, uint64_t mops_cap, float* zoops, uint64_t* z) {
    if (!c || !s || !v || !vbg || !zoops || !z) { set_error("bamm_logodds: null argument"); return BAMM_ERR_ARG; }
    if (K > BAMM_MAX_ORDER || W == 0) { set_error("bamm_logodds: bad K/W"); return BAMM_ERR_ARG; }
    if (s->n && s->min_len < W) { set_error("a sequence is shorter than the motif (W=%u)", W); return BAMM_ERR_ARG; }
    if (s->n == 0) return BAMM_OK;
    BAMM_HIP(hipSetDevice(c->device));
    const uint32_t Y = (uint32_t)ipow4(K + 1), Ys = Y + 1, Kbg = std::min(bg_order, K);
    const uint32_t Yb = (uint32_t)ipow4(Kbg + 1);
    // Motif::calculateLogS (Motif.cpp:471-483) with the host's logf, laid out [j][y] + neutral row
    std::vector<float> tab((size_t)W * Ys, 0.0f);
    const float* vK = v + v_offset(K, W);
    const float* b = vbg + bg_offset(Kbg);
    for (uint32_t y = 0; y < Y; y++)
        for (uint32_t j = 0; j < W; j++)
            tab[(size_t)j * Ys + y] = logf(vK[(size_t)y * W + j] + 1e-5f) - logf(b[y % Yb]);
    std::vector<uint64_t> moff(s->n + 1, 0);
    for (uint64_t n = 0; n < s->n; n++) moff[n + 1] = moff[n] + (s->h_len[n] - W + 1);
    if (mops && mops_cap < moff[s->n]) { set_error("mops buffer too small"); return BAMM_ERR_ARG; }
    hipStream_t st = c->stream;
    ExcK* exc = nullptr;
    int rc = exceptions_for_order(s, K, &exc);
    if (rc) return rc;
    float *d_tab = nullptr, *d_mops = nullptr, *d_zoops = nullptr;
    uint64_t* d_moff = nullptr;
    uint32_t* d_z = nullptr;
    uint8_t* d_smask = nullptr;
    auto cleanup = [&]() { (void)hipFree(d_tab); (void)hipFree(d_mops); (void)hipFree(d_zoops); (void)hipFree(d_moff); (void)hipFree(d_z); (void)hipFree(d_smask); };
    if ((rc = dev_upload(c, &d_tab, tab.data(), tab.size())) || (rc = dev_upload(c, &d_moff, moff.data(), moff.size())) ||
        (rc = dev_alloc(&d_zoops, s->n)) || (rc = dev_alloc(&d_z, s->n)) || (mops && (rc = dev_alloc(&d_mops, moff[s->n]))) ||
        (seq_mask && (rc = dev_upload(c, &d_smask, seq_mask, s->n)))) {
        cleanup();
        return rc;
    }
    if (seq_mask) {                                          // sequences outside the subset report zeros
        hipError_t e = hipMemsetAsync(d_zoops, 0, s->n * sizeof(float), st);
        if (e == hipSuccess) e = hipMemsetAsync(d_z, 0, s->n * sizeof(uint32_t), st);
        if (e == hipSuccess && mops) e = hipMemsetAsync(d_mops, 0, moff[s->n] * sizeof(float), st);
        if (e != hipSuccess) { set_error("hipMemsetAsync failed: %s", hipGetErrorString(e)); cleanup(); return BAMM_ERR_HIP; }
    }
    for (size_t bi = 0; bi < s->buckets.size() && !rc; bi++) {
        const Bucket& bk = s->buckets[bi];
        ScoreKernelArgs a{};
        a.sv = make_view(s, exc, bk, d_smask);
        a.K = K; a.W = W; a.Y = Y; a.s = d_tab; a.mops = d_mops; a.mops_off = d_moff; a.zoops = d_zoops; a.z = d_z;
        // beyond the length classes, or a log-odds table beyond the LDS of a CU (orders >= 6 at usual widths): the
        // window-by-window scorer reads the table from global memory, same sums in the same order
        if (bk.mclass == kLongClass || (size_t)W * Ys * sizeof(float) > 160u * 1024u) {
            rc = launch_long_score(a, std::min(bk.count, (uint32_t)std::max(1, c->num_cus) * 8u), st);
            continue;
        }
        const uint32_t threads = default_threads(c, bk.mclass);
        uint32_t blocks = default_blocks(c, threads);
        blocks = std::max(1u, std::min(blocks, (bk.count + threads / 64u - 1) / (threads / 64u)));
        rc = launch_score(bk.mclass, a, blocks, threads, st);
    }
    std::vector<uint32_t> hz(s->n);
    if (!rc) {
        hipError_t e = ctx_download(c, zoops, d_zoops, s->n * sizeof(float)) ? hipErrorUnknown : hipSuccess;
        if (e == hipSuccess && ctx_download(c, hz.data(), d_z, s->n * sizeof(uint32_t))) e = hipErrorUnknown;
        if (e == hipSuccess && mops && ctx_download(c, mops, d_mops, moff[s->n] * sizeof(float))) e = hipErrorUnknown;
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) { set_error("bamm_logodds: copy failed: %s", hipGetErrorString(e)); rc = BAMM_ERR_HIP; }
    }
    if (!rc) for (uint64_t n = 0; n < s->n; n++) z[n] = hz[n];
    cleanup();
    return rc;
}

}  // extern "C"
