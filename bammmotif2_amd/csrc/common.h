// Shared declarations of the gfx950 EM hot path (host side of the C ABI + kernel launchers).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "../../include/bamm_em.h"

namespace bamm {

void set_error(const char* fmt, ...);

#define BAMM_HIP(expr)                                                                        \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) {                                                               \
            ::bamm::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, \
                              __LINE__);                                                      \
            return BAMM_ERR_HIP;                                                              \
        }                                                                                     \
    } while (0)

// raise a kernel's dynamic-LDS limit above the 64 KiB default; a refusal is reported here with the
// size that was asked for, not later as a generic launch error
inline int allow_lds(const void* kernel, size_t lds) {
    if (lds <= 64 * 1024) return BAMM_OK;
    const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
        set_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize = %zu) failed: %s", lds, hipGetErrorString(e));
        return BAMM_ERR_HIP;
    }
    return BAMM_OK;
}

// A launcher called with this block count makes the kernel ready to launch -- the runtime loads the code object of the
// kernel's translation unit on first use: 0.8 ms for the mixed-row kernels, 8-14 ms for kernels.hip -- and launches
// nothing.  bamm_em_create primes the kernels of the handle's passes, so that the first pass costs what the others do.
constexpr uint32_t kPrimeOnly = 0xffffffffu;
inline int prime_kernel(const void* kernel) {
    hipFuncAttributes at;
    const hipError_t e = hipFuncGetAttributes(&at, kernel);
    if (e != hipSuccess) { set_error("hipFuncGetAttributes failed: %s", hipGetErrorString(e)); return BAMM_ERR_HIP; }
    return BAMM_OK;
}

inline size_t ipow4(size_t e) { return size_t(1) << (2 * e); }
inline size_t v_offset(size_t k, size_t W) { return W * ((ipow4(k + 1) - 4) / 3); }
inline size_t v_size(size_t K, size_t W) { return v_offset(K + 1, W); }
inline size_t bg_offset(size_t k) { return (ipow4(k + 1) - 4) / 3; }
inline size_t bg_size(size_t K) { return bg_offset(K + 1); }

// internals shared between abi.cpp and comm.cpp
int ctx_device(const bamm_ctx* c);
hipStream_t ctx_stream(const bamm_ctx* c);
int comm_allreduce_i64(bamm_comm* c, void* dev_ptr, size_t n_words, hipStream_t st);   // ncclAllReduce(ncclInt64, ncclSum)
bamm_ctx* comm_ctx(const bamm_comm* c);
bool comm_aborted(const bamm_comm* c);      // bamm_comm_abort was called on it (by any thread)
struct PeerArgs;
int comm_peer_setup(bamm_comm* c, uint32_t stride_words, int* ready);   // collective; *ready = 0 is a refusal, not an error
const char* comm_peer_why(const bamm_comm* c);
void comm_peer_args(const bamm_comm* c, PeerArgs* p);
unsigned long long comm_peer_next_seq(bamm_comm* c);

// positions-per-lane classes the sequence kernels are instantiated for (L <= 64*M)
constexpr int kNumMClasses = 23;
extern const int kMClasses[kNumMClasses];
int m_class_for_len(uint32_t L);  // index into kMClasses, or -1 when L > 64*128 (kLongClass: long_seq.hip)
constexpr int kLongClass = -1;

// ---------------------------------------------------------------- device views ----------
struct SeqView {                 // one length bucket of a resident sequence set
    const uint32_t* words;       // 2-bit stream, 16 positions / word, big-endian in word
    const uint64_t* word_off;    // [N+1]
    const uint32_t* len;         // [N]
    const uint64_t* pos_off;     // [N+1] prefix sum of len (layout of r)
    const uint64_t* exc_off;     // [N+1] exceptions relevant at this order
    const uint2*    exc;         // (position, y value)
    const uint8_t*  mask;        // nullable
    const uint32_t* idx;         // nullable: sequence ids of this bucket
    uint32_t        count;       // sequences in this bucket
};

struct EmKernelArgs {
    SeqView  sv;
    uint32_t K, W, Y;            // Y = 4^(K+1)
    uint32_t logC;               // log2 of the private count-table copies per block
    uint32_t sparse_cap;         // max non-zero windows handled by the sparse M-step (0 = always dense)
    uint32_t sparse_wave_bytes;  // per-wave LDS scratch of the sparse M-step
    const float* s;              // device, [W][Y+1], last row entry = neutral element
    const float* q;              // device scalar
    long long* acc;              // nullable.  The pass's ONE fused accumulator [Y*W | llh | sum_r | n_seqs] of 64-bit
                                 // integers (counts in [y][j] order, fixed point): every block adds its table with
                                 // no-return device-scope atomics.  Integer sums are exact and order-free.
    float    fix_scale;          // r is multiplied by this power of two before the 2^40 conversion (1 = 2^-40 units)
    float*   r_out;              // WRITE_R: reference layout, r_base subtracted; sliced path: slot-indexed state / r
    uint64_t r_base;             // pos_off of the first requested sequence
    uint32_t seq_begin, seq_end; // WRITE_R range filter (sequence ids)
    // column-sliced path with the whole odds table in LDS: the E pass leaves, per sequence, the compacted list of
    // its windows with a non-zero fixed-point addend instead of all L responsibilities (list_r != nullptr)
    float*    list_r;            // [pos_off[seq] + k] responsibility of the k-th listed window
    uint16_t* list_p;            // [pos_off[seq] + k] its slot: the position of the window's last column
    uint32_t* list_n;            // [seq] listed windows
    // bamm_em_optimize() enqueues one pass ahead of the pass whose (llh, v_diff) it is waiting for; k_update sets
    // this word when the stop rule (EM.cpp:117-118) fires, and a pass enqueued behind it does nothing
    const uint32_t* stop;        // nullable
    // sliced path, choice per pass between compacted lists and dense r, made ON THE DEVICE from the number of non-zero
    // windows the previous pass's E kernel counted (while the model is uninformative a list holds every window and a
    // list pass costs 26 ms against 15 ms for the dense walk, config 4): both flavours of a pass are enqueued, each
    // launch runs only if (*nnz_prev > nnz_limit) == (run_if_long != 0)
    const unsigned long long* nnz_prev;   // nullable: always run
    unsigned long long nnz_limit;
    uint32_t run_if_long;
    unsigned long long* nnz_out;          // nullable: the E pass adds its count of windows with a non-zero fixed-point addend
};

// ---- grouped-column kernel (grouped.hip): G motif columns (K+G = 4 or 5) share one table row ------
// Row index of a position p = the (K+G)-mer ending at p; rows beyond the full ones cover groups cut
// by the EM.cpp:167 truncation (p >= L-W+1), one neutral row, and per-wave "virtual" rows that
// stand in for group ends next to an N exception (Sequence.cpp:38).
struct GrpGeom {
    uint32_t G, T, Tq, delta;    // group size, groups = ceil(W/G), quads of groups, G*T - W neutral front columns
    uint32_t Ts;                 // cells per (row, copy) of the count table: T, or T | 1 (layout bit 2)
    uint32_t Rf;                 // full rows = 4^(K+G)
    uint32_t layout, np;         // see grp_geometry(); partial classes of table rows (0 or G-1)
    uint32_t base[3], psize[3];  // partial class d (d+1 trailing positions neutral): first row, rows = 4^(K+G-1-d)
    uint32_t Rn, R0, Bj, Bv, Rtot;   // neutral row, first virtual row, virtual rows per wave for exceptions / in all, rows
    uint32_t rowstride;          // floats per row of the odds table [Rtot][Tq | 1][4] (odd number of quads)
    uint32_t off_sg, off_s1, off_stat, off_ng, off_n1, off_wave, wave_bytes;   // LDS byte offsets
    uint32_t cap;                // K = 3 kernels: 1 = the single-column table is in LDS, 0 = read from global memory
    uint32_t lds_bytes;
    // layout bit 3, mixed rows (mixed_kernel.h, K = 2): mixB groups of 3 columns on 5-mer rows (the fields above
    // describe that narrow table; G = 4 is the widest group), then mixA groups of 4 columns on 6-mer rows
    uint32_t mixA, mixB, off_sg6, off_ng6;
};

// ---- the pass's all-reduce in the tail of the sequence kernel, over peer-mapped memory (update_kernel.h:
// peer_allreduce_tail; comm.cpp: comm_peer_setup) ----
// Every rank owns an INBOX in its own HBM, fine-grained, mapped into every peer: [3 slots][world sources][stride] entries
// of 16 bytes, an entry = one int64 word as two 8-byte halves {32 bits of data | the pass's 32-bit sequence number}.
// The last block of the pass's launch to finish reads the GPU's totals back from the accumulator, stores them into its
// buffer of every peer's inbox (system-scope 8-byte stores over xGMI), collects the peers' entries from its own inbox
// as they arrive (a half is valid when it carries the sequence number: no fence, no separate flag; every poll is bounded
// by a wall-clock deadline) and leaves totals + peers in the accumulator: integer sums, the same on every rank.  The
// kernel boundary hands the all-reduced accumulator to what follows -- no collective launch, no host code.
constexpr uint32_t kPeerMaxWorld = 8;
struct PeerArgs {
    uint32_t world, rank;        // world <= 1: off
    uint32_t stride;             // entries per (slot, source) buffer
    uint32_t words;              // words all-reduced: Y*W + 3
    uint32_t slot;               // seq % 3
    unsigned long long seq;      // this pass's sequence number (the ranks count in step; its low 32 bits tag the entries; never 0)
    unsigned long long timeout_ticks;   // of the 100 MHz wall clock
    void* inbox;                 // this rank's
    void* peer[kPeerMaxWorld];   // peer[d]: rank d's inbox as mapped on this device (peer[rank] unused)
    uint32_t* ticket;            // a zeroed device word: the blocks of the launch draw tickets, the last one all-reduces and resets it
    uint32_t* err;               // device word: 0 = fine, else 1 + the rank whose entries did not arrive in time; a non-zero
                                 // word makes every later launch of the handle a no-op (the host turns it into BAMM_ERR_COMM)
};

struct UpdateArgs {
    uint32_t K, W, Kbg;          // Kbg = min(bg_order, K)
    long long* acc;              // [Y*W + 3]: n_K in [y][j] (units of count_unit), llh, sum_r (fixed point), n_seqs;
                                 // k_update: consumed AND zeroed; fused into the next pass's kernel: read only
    long long* acc_zero;         // nullable: another slot of the accumulator ring to clear (the one the pass after next adds
                                 // into; k_update: the slot the last fused kernel read and could not clear itself)
    double   count_unit;         // value of one unit of the counts (2^-40 unless the set is huge)
    const float* vbg;            // flat bg conditionals (orders 0..bg_order)
    const float* A;              // [K+1][W]
    float* n;                    // flat counts (all orders)
    float* v;                    // flat conditionals (all orders): the new model
    const float* v_old;          // nullable: the model the pass ran with, when it is not `v` itself (fused updates keep the
                                 // two apart: every block reads the old one while the writer block stores the new one)
    float* s;                    // [W][Y+1] linear odds for the next E-step
    float* q;                    // device scalar (input)
    float* q_out;                // device scalar for the next pass (may alias q)
    float* status;               // [8]: llh, v_diff, q, iteration, ...
    float* trace;                // [cap][3]
    uint32_t trace_cap;
    uint32_t* iteration;         // device counter
    int32_t optimize_q;          // re-estimate q in this pass (EM.cpp:99: the first five passes of an optimize() call)
    double n_seqs_override;      // >0: use instead of red[..+2]
    // optimize(): the stop rule evaluated where its inputs are (EM.cpp:117-118, same float comparisons as the host's)
    uint32_t* stop;              // nullable; set to 1 when the rule fires; a non-zero word makes this launch a no-op
    float epsilon;               // EM.h:62
    float llh_prev;              // likelihood before this pass when it is the call's first ...
    int32_t llh_prev_from_status;   // ... else *llh_in, as the previous update left it
    const float* llh_in;         // the previous update's log-likelihood (a slot the writer does not store into)
    float* llh_out;              // nullable: where this update leaves its own
    uint32_t opt_iteration;      // 1-based pass number inside this optimize() call (`iteration > 10`)
    unsigned long long* status_mirror;   // nullable: pinned host memory (device address) that receives the first six status words
                                 // as well, each as a self-validating 8-byte word {opt_iteration | float bits}: the host
                                 // polls them, no copy and no event in the stream
    // tables beyond the LDS form of the update (K >= 3 at usual widths): the update is spread over blocks in two
    // launches (k_update_counts, k_update_model) instead of one block walking up to millions of cells
    double*   partial;           // nullable ([kUpdateMaxBlocks]): per-block partial sums of v_diff, summed in block order by the last block
    uint32_t* ticket;            // a zeroed word: the blocks of k_update_model draw tickets from it, the last one resets it
};
constexpr uint32_t kUpdateMaxBlocks = 1024;

// the update with every order of n staged in LDS (update_kernel.h): scratch bytes, and
// whether it applies -- at most 2048 cells of the top order (two per thread at 1024 threads), tables within 60 KiB
inline size_t update_lds_bytes(uint32_t K, uint32_t W) {   // n of all orders, A, vbg (each padded to 8 bytes), 16 + 4 doubles
    auto even = [](size_t x) { return (x + 1) & ~size_t(1); };
    return (even(v_size(K, W)) + even((size_t)(K + 1) * W) + even(bg_size(K))) * sizeof(float) + (16 + 4) * sizeof(double);
}
inline bool update_fits_lds(uint32_t K, uint32_t W) { return ipow4(K + 1) * W <= 2048u && update_lds_bytes(K, W) <= 60u * 1024u; }

struct GrpKernelArgs {
    EmKernelArgs e;
    GrpGeom g;
    const uint4* xrec;           // per sequence: x = lo | B<<12 (group ends lo..lo+B-1 need a virtual row), y/z/w = exact y of
                                 // the positions lo-G+1.., ONE bit string of 7-bit fields (10-bit at K = 3; the value Y =
                                 // position before the sequence)
    // K = 3, accumulating pass: the non-zero sums the fix lanes took out of their virtual count rows, logged per
    // wave (8-byte entries: grouped_kernel.h, GrpLogEntry) and folded into single-column bins in the block epilogue
    unsigned long long* fix_log;
    uint32_t fix_log_cap;        // entries per wave: sequences of the wave x fix lanes (Bv * T)
    // fused update (update_kernel.h): the previous pass's model update runs in this launch's block prologue instead of
    // a kernel of its own; e.s / e.q are then what the writer block publishes (later launches of the pass, getR)
    uint32_t fused;              // 0 = e.s / e.q hold the model as usual
    uint32_t upd_off;            // LDS byte offset of the update's scratch (update_lds_bytes), inside the count tables
    float*   s_block;            // [blocks][W * (Y + 1)]: every block's copy of the odds table for its fix lanes
    UpdateArgs upd;
    PeerArgs peer;               // in-kernel all-reduce (world <= 1: off)
};

// geometry for (K, W) with `waves` waves per block, M positions per lane; false when the kernel does not apply
bool grp_geometry(uint32_t K, uint32_t W, uint32_t G, int M, uint32_t waves, bool accum, uint32_t logC, uint32_t layout,
                  GrpGeom* out);
// enough_work: the launch has enough sequences for the mixed rows' larger tables to pay (their prologue and
// epilogue cost 4-8 us more per launch: break-even near 40k sequences of 200 bp)
bool grp_plan(uint32_t K, uint32_t W, int M, uint32_t waves, bool many_exceptions, bool enough_work, uint32_t forced_G,
              int forced_layout, uint32_t* G, uint32_t* logC, uint32_t* layout);   // false: use k_em_seq
bool grp_supported_class(int M, uint32_t K);
int launch_em_grp_long(int mclass, bool accum, bool write_r, const GrpKernelArgs& a, uint32_t blocks, uint32_t threads,
                       hipStream_t st);   // grouped_long.hip: 20..32 positions per lane
int launch_em_grp_xl(int mclass, bool accum, bool write_r, const GrpKernelArgs& a, uint32_t blocks, uint32_t threads,
                     hipStream_t st);     // grouped_xl.hip: 40 / 48 positions per lane
uint32_t grp_max_threads(int M);   // block size the grouped kernel of this length class is built for
// mixed rows: geometry (false: does not apply / does not fit) and the launchers of its two translation units
bool mix_geometry(uint32_t K, uint32_t W, int M, uint32_t waves, bool accum, GrpGeom* out);
int launch_em_mix(int mclass, bool accum, bool write_r, const GrpKernelArgs& a, uint32_t blocks, uint32_t threads, hipStream_t st);
int launch_em_mix1(int mclass, bool accum, bool write_r, const GrpKernelArgs& a, uint32_t blocks, uint32_t threads, hipStream_t st);
int launch_em_grp(int mclass, bool accum, bool write_r, const GrpKernelArgs& a, uint32_t blocks, uint32_t threads,
                  hipStream_t st);

struct SeedKernelArgs {          // Motif::initFromPWM's pass over the sequences (seed.hip)
    SeqView  sv;                 // every resident sequence, exceptions of order K
    uint32_t K, W, Y;            // Y = 4^(K+1)
    uint32_t max_len, vsize;     // longest sequence, cells of the count table (all orders)
    uint32_t table_bytes, count_bytes, wave_bytes;   // LDS layout, filled in by the launcher
    const float* score;          // [4][W] floored PWM / 0th-order background (Motif.cpp:205-226)
    float    q;
    const double* u;             // [N] the uniform variate of each sequence's draw
    int*     counts;             // [vsize] flat [k][y][j], zeroed by the caller
    uint32_t* z_out;             // nullable: sampled index per sequence (0 = no motif)
    unsigned char* wave_scratch; // per-wave arrays in global memory (sequences whose arrays do not fit the LDS)
};
// bytes of global scratch the launch needs for its per-wave arrays (0: they fit the LDS), and the launch itself
size_t seed_global_scratch_bytes(const SeedKernelArgs& a, uint32_t num_cus);
int launch_seed_pwm(SeedKernelArgs a, uint32_t num_cus, hipStream_t st);

struct ScoreKernelArgs {
    SeqView  sv;
    uint32_t K, W, Y;
    const float* s;              // device log-odds table [W][Y+1], pad = 0
    float*   mops;               // nullable, concatenated L-W+1 per sequence
    const uint64_t* mops_off;    // [N+1]
    float*   zoops;              // [N]
    uint32_t* z;                 // [N]
};


struct MaskSelect {              // device state of the radix select (EM.cpp:329-343)
    double   pos_count;          // number of windows (all ranks)
    double   rank;               // remaining 0-based rank in the descending order
    unsigned long long listed;   // windows at or above the cut-off (this rank)
    uint32_t prefix;             // bit-pattern prefix chosen so far
    float    cutoff;
};

struct MaskKernelArgs {          // EM::mask kernels (mask.hip)
    SeqView  sv;                 // every resident sequence, exceptions of order K
    uint32_t K, W, Y;
    uint32_t max_len;            // sizes the per-wave LDS arrays
    uint32_t wave_bytes;         // mask_wave_bytes(max_len, arrays in the global scratch)
    uint32_t table_bytes;        // block-shared table in front of the per-wave arrays
    const float* v0;             // order-0 conditionals [4][W]
    const float* vbg0;           // order-0 background [4]
    const float* s;              // [W][Y+1] linear odds of the current model
    float*   q;                  // device scalar
    float*   q_seq;              // nullable: q each sequence saw in the order-0 pass (optimizeQ)
    float    n_total;            // number of training sequences (optimize_q's N)
    float    fix_scale;          // as EmKernelArgs::fix_scale
    float*   r;                  // responsibilities, reference layout (pos_off[n] + r-index)
    uint32_t* bits;              // 1 bit per r slot: window belongs to the top-f set
    long long* hist;             // [2049] window counts per bin + the number of windows: integers, so that the
                                 // all-reduce across ranks is the same int64 sum as the EM pass's
    MaskSelect* sel;
    unsigned long long* partial_n;
    double*  partial_stat;
    uint32_t j0, j1;             // column range of k_mask_m
    unsigned char* wave_scratch; // nullable: the per-wave arrays live here, one region per wave of the grid, instead of
                                 // in LDS (sequences whose arrays do not fit beside the block's table); 32-bit lists there
    long long* acc_direct;       // nullable: k_mask_m adds straight into the pass's accumulator ([y][j] + statistics) instead
                                 // of a per-block table in LDS (orders whose count column exceeds the LDS)
};

// launchers (mask.hip)
size_t mask_wave_bytes(uint32_t max_len, bool wide_lists);   // wide: 32-bit window lists (arrays in the global scratch)
int launch_mask_init(const MaskKernelArgs& a, bool serial, uint32_t blocks, uint32_t threads, hipStream_t st);
int launch_mask_hist(const MaskKernelArgs& a, int pass, uint32_t blocks, hipStream_t st);
int launch_mask_pick(const MaskKernelArgs& a, int pass, float f, hipStream_t st);
int launch_mask_bits(const MaskKernelArgs& a, uint32_t blocks, hipStream_t st);
int launch_mask_e(const MaskKernelArgs& a, bool s_in_lds, uint32_t blocks, uint32_t threads, hipStream_t st);
int launch_mask_m(const MaskKernelArgs& a, uint32_t blocks, uint32_t threads, hipStream_t st);

// launchers (kernels.hip)
size_t em_lds_bytes(uint32_t W, uint32_t Y, bool accum, uint32_t logC, size_t scratch);
uint32_t pick_log_copies(uint32_t W, uint32_t Y, uint32_t blocks_per_cu, size_t scratch);
uint32_t sparse_cap_for(int M);
size_t sparse_wave_bytes(int M);
int launch_em_seq(int mclass, bool accum, bool write_r, const EmKernelArgs& a, uint32_t blocks,
                  uint32_t threads, hipStream_t st);
size_t e_slice_lds_bytes(uint32_t cols, uint32_t Y);
size_t m_slice_lds_bytes(uint32_t cols, uint32_t Y, uint32_t logC);
int launch_e_slice(int mclass, const EmKernelArgs& a, uint32_t j0, uint32_t j1, bool last, uint32_t blocks,
                   uint32_t threads, hipStream_t st);
size_t m_slice_wave_bytes(int M, uint32_t cap);
int launch_m_slice(int mclass, const EmKernelArgs& a, uint32_t j0, uint32_t j1, bool r_reversed, uint32_t blocks,
                   uint32_t threads, hipStream_t st);
int launch_m_list(int mclass, const EmKernelArgs& a, uint32_t j0, uint32_t j1, uint32_t blocks, uint32_t threads, hipStream_t st);
size_t m_list_lds_bytes(uint32_t cols, uint32_t Y, uint32_t logC, int M, uint32_t waves);
// sequences beyond the length classes (long_seq.hip): one workgroup per sequence, window by window
int launch_long_em(const EmKernelArgs& a, bool accum, bool write_r, bool slot_layout, uint32_t blocks, hipStream_t st);
int launch_long_score(const ScoreKernelArgs& a, uint32_t blocks, hipStream_t st);
int launch_score(int mclass, const ScoreKernelArgs& a, uint32_t blocks, uint32_t threads,
                 hipStream_t st);
// EM::mask only: its kernels still leave one partial table per block; summed into the fused accumulator
int launch_reduce_partials(const unsigned long long* partial_n, const double* partial_stat, uint32_t blocks,
                           uint32_t W, uint32_t Y, long long* acc, hipStream_t st);
int launch_make_s(const float* v, const float* vbg, uint32_t K, uint32_t W, uint32_t Kbg, float* s,
                  hipStream_t st);
int launch_update(const UpdateArgs& a, hipStream_t st);
int prime_model_kernels();
int launch_stat_only(long long* acc, uint32_t cells, float* status, hipStream_t st);
uint32_t max_threads_for_mclass(int mclass);

}  // namespace bamm
