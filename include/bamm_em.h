/* bamm_em.h -- C ABI of the MI355X (gfx950) EM-refinement hot path of BaMMmotif2.
 *
 * The reference has no FFI layer: the seam is the C++ class `EM` (src/refinement/EM.h:11-69)
 * and `ScoreSeqSet` (src/seq_scoring/ScoreSeqSet.h:14-53), which read
 * `Sequence::getKmer()/getL()` (src/init/Sequence.h:26,36), `Motif::getV/getA/getK/getW/getQ`
 * (src/init/Motif.h:30-38) and `BackgroundModel::getV/getOrder` (src/init/BackgroundModel.h:34-35)
 * and mutate the caller's Motif in place.  This header is what a binding for that seam binds:
 * plain pointers and sizes, no C++ / torch types.  INTEGRATION.md shows the reference-side
 * glue (an `EM` whose methods forward to these entry points).
 *
 * Conventions
 *   - every function returns BAMM_OK (0) or a negative error code; bamm_last_error() gives
 *     the message of the last failure on the calling thread.  Nothing calls exit().
 *     (the reference only ever `exit(1)`s after a std::cerr line, e.g. Global.cpp:127-131.)
 *   - all host buffers are caller-owned and only read/written during the call.
 *   - handles are not thread-safe individually; distinct handles may be used from distinct
 *     threads (FDR.cpp:37 runs folds concurrently).
 *   - flat tensor layouts (identical to oracle/bamm_oracle.h):
 *       v, n, p : [k][y][j] row-major, orders concatenated, order k at W*(4^(k+1)-4)/3
 *       vbg     : [k][y], order k at (4^(k+1)-4)/3
 *       A       : [k][j], (K+1) x W          (Motif.cpp:43-46)
 *       s       : [y][j], 4^(K+1) x W
 *       r       : per sequence L floats, reference's reversed index: r[L-W-i] <-> window
 *                 start i, slots >= L-W+1 are zero (EM.cpp:173,190-192)
 */
#ifndef BAMM_EM_H_
#define BAMM_EM_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BAMM_OK              0
#define BAMM_ERR_ARG        -1   /* bad argument (null, size, order > BAMM_MAX_ORDER, L < W ...) */
#define BAMM_ERR_HIP        -2   /* HIP runtime failure (message carries hipGetErrorString)      */
#define BAMM_ERR_NO_DEVICE  -3   /* no gfx950 device / extension built for another arch         */
#define BAMM_ERR_UNSUPPORTED -4  /* shape outside the kernels' envelope         */
#define BAMM_ERR_STATE      -5   /* call order (e.g. MStep before any EStep)                    */
#define BAMM_ERR_COMM       -6   /* the caller's all-reduce callback failed                     */

#define BAMM_MAX_ORDER      10   /* kmer_ spans 11 bases (Sequence.cpp:37)                      */
#define BAMM_MAX_SEQ_POSITIONS 8192u /* per sequence (reverse strand and separator included): up to
                                      * here a sequence lives in the registers of one wavefront; longer
                                      * ones are accepted and walked window by window (long_seq.hip)  */

typedef struct bamm_ctx  bamm_ctx;   /* one device + one stream                                   */
typedef struct bamm_seqs bamm_seqs;  /* a sequence set resident in HBM (2-bit packed)             */
typedef struct bamm_em   bamm_em;    /* one EM run; replaces `class EM` (EM.h:11-69)              */
typedef struct bamm_comm bamm_comm;  /* one rank of an RCCL communicator, bound to a context       */

const char* bamm_last_error(void);
const char* bamm_version(void);

/* ------------------------------------------------------------------ packing (host only) --
 * Replaces what EM/ScoreSeqSet read through Sequence::getKmer() (EM.cpp:152, :233,
 * ScoreSeqSet.cpp:44): `kmer_[i]` = up to 11 bases ending at i, newest base least
 * significant (Sequence.cpp:35-41).  The packed form keeps 2 bits per position (digit 0 of
 * kmer_[i]) in big-endian order inside 32-bit words, 16 positions per word, every sequence
 * starting on a word boundary, plus an exception list for the positions whose 11-mer cannot
 * be rebuilt from those bits (the reference randomises an unknown base independently per
 * (position, digit), Sequence.cpp:38, and lets the byte 'N' leak into reverse complements,
 * Alphabet.cpp:50).  Pure CPU code: usable and testable without a GPU.                       */
typedef struct bamm_packed {
    uint64_t  n_seqs;
    uint64_t  n_words;
    uint64_t  n_exc;
    uint64_t  total_len;     /* sum of L                                                     */
    uint32_t  max_len;
    uint32_t  min_len;
    uint32_t* words;         /* [n_words]                                                    */
    uint64_t* word_off;      /* [n_seqs+1] first word of each sequence                       */
    uint32_t* len;           /* [n_seqs]   L                                                 */
    uint64_t* exc_off;       /* [n_seqs+1]                                                   */
    uint32_t* exc_pos;       /* [n_exc] position inside its sequence, ascending              */
    uint32_t* exc_kmer;      /* [n_exc] kmer_[pos] mod 4^11 as the reference holds it        */
    uint32_t* exc_clean;     /* [n_exc] what the 2-bit stream alone would give               */
} bamm_packed;

/* kmer: concatenated kmer_ arrays, off[n_seqs+1] (positions).  Values are taken mod 4^11.    */
int  bamm_pack_kmers(const uint64_t* kmer, const uint64_t* off, uint64_t n_seqs, bamm_packed** out);
/* same input as the reference holds it: one pointer per Sequence (getKmer()) + getL()        */
int  bamm_pack_kmer_ptrs(const uint64_t* const* kmer_ptrs, const uint64_t* L, uint64_t n_seqs,
                         bamm_packed** out);
/* from the alphabet codes of FASTA records (0 = N, 1..4 = A,C,G,T; off[n_seqs+1]) -- restates
 * Sequence::Sequence (Sequence.cpp:4-43,91-99): appends the reverse complement unless
 * single_strand and draws libc rand()%4 for unknown bases in the reference's order, so after
 * the caller's srand(42) (mainBaMM.cpp:22) the result equals packing the reference's kmer_.   */
int  bamm_pack_codes(const uint8_t* codes, const uint64_t* off, uint64_t n_seqs, int single_strand,
                     bamm_packed** out);
/* the same when the libc stream starts at srand(seed) (the reference seeds once, mainBaMM.cpp:22, and its positives
 * are read first): glibc's generator is restated and checked against the running libc, so the draws are taken on all
 * host threads (each jumps to its share of the one stream) instead of one after the other; falls back to
 * srand(seed) + rand() where the check fails.  Same packing as srand(seed); bamm_pack_codes(...).  Postcondition on
 * either path: libc's stream stands at srand(seed), NOT advanced past the draws (the reference's later consumers all
 * reseed: SeqGenerator.cpp:35, FDR.cpp:153); a caller that continues the one stream behind the positives uses
 * bamm_pack_codes, which draws from libc's rand() itself.                                                       */
int  bamm_pack_codes_seeded(const uint8_t* codes, const uint64_t* off, uint64_t n_seqs, int single_strand,
                            uint32_t seed, bamm_packed** out);
/* host threads the packing helpers may use (0 = a default of at most 8); the result never depends
 * on it: the rand() draws are taken serially, in the reference's order                        */
void bamm_set_host_threads(uint32_t n);
/* inverse (host): rebuild kmer_[i] mod 4^(K+1) for every position -- used by tests          */
int  bamm_unpack_y(const bamm_packed* p, uint32_t K, uint32_t* y_out /* [total_len] */);
void bamm_packed_free(bamm_packed* p);

/* contiguous shard [begin,end) of rank `rank` of `world`, balanced by sum(L-W+1) */
int  bamm_shard_range(const uint32_t* len, uint64_t n_seqs, uint32_t W, uint32_t rank,
                      uint32_t world, uint64_t* begin, uint64_t* end);

/* ------------------------------------------------------------------ context ------------- */
/* stream: a hipStream_t created by the caller (e.g. torch's current stream), or NULL to let
 * the context create its own.  Fails with BAMM_ERR_NO_DEVICE when no gfx950 GPU is visible. */
int  bamm_ctx_create(int device, void* hip_stream, bamm_ctx** out);
int  bamm_ctx_destroy(bamm_ctx* ctx);
int  bamm_ctx_sync(bamm_ctx* ctx);
int  bamm_ctx_device_name(bamm_ctx* ctx, char* buf, size_t cap);
/* launch geometry of the sequence kernels (0 = default); exposed for tuning/benchmarks      */
int  bamm_ctx_set_launch(bamm_ctx* ctx, uint32_t blocks, uint32_t threads_per_block);

/* kernel-selection switches, applied to EM handles created afterwards (benchmarks and the tests that
 * compare one kernel with another; results agree within the parity bar whatever is chosen):
 *   "grouped"      1/0  grouped-column kernel for K <= 3 (default 1; 0 = one column at a time)
 *   "group_size"   0 = planner's choice, 2..4 = columns per table row
 *   "group_layout" -1 = planner's choice, 0..3 = table layout of the uniform rows (csrc/grouped.hip:
 *                       grp_geometry), 8 = mixed rows only (csrc/mixed_kernel.h; where they do not apply
 *                       the sequences go one column at a time)
 *   "sparse"       1/0  compacted lists of the non-zero windows in the M-step (default 1)
 *   "e_fused"      1/0  sliced path: whole-table E pass when the odds table fits LDS (default 1)
 *   "e_list"       1/0  sliced path: the E pass hands the M slices compacted lists of the non-zero
 *                       windows instead of all responsibilities (default 1)
 *   "adaptive_lists" 1/0  sliced path: per pass, lists or dense r between the E pass and the M slices, chosen on the
 *                       device from the previous pass's count of non-zero windows (default 1; 0 = always lists)
 *   "list_threshold_pct" 0..100  ... lists when fewer than this percentage of the windows was non-zero (default 45)
 *   "fused_update" 1/0 inside iterate() / optimize() the model update of pass p runs in the block prologue of
 *                       pass p+1's first kernel instead of a launch of its own (default 1; K <= 2-sized tables)
 *   "update_blocks" 1/0 tables beyond the update's LDS form (k >= 3 at usual widths): the model update spread over
 *                       blocks in three short launches instead of one block (default 1; same model bits)
 *   "scratch_cache_mb" n  idle set-sized scratch blocks (dense r, lists, logs) a context keeps for its next handle
 *                       instead of freeing them (default: a quarter of the device's memory; 0 = keep nothing)
 *   "scratch_poison" 1/0 tests: fill every such block with 0xFF bytes when it is handed out (default 0)
 * There are no environment variables that change what the library computes or launches.          */
int  bamm_ctx_set_tuning(bamm_ctx* ctx, const char* key, int value);

/* ------------------------------------------------------------------ sequences ----------- */
/* Uploads sequences [begin,end) of `p`; they stay resident and are shared (ref-counted) by
 * any number of EM handles -- CV folds pass a mask instead of copying (FDR.cpp:49-57).
 * Any length: sequences up to BAMM_MAX_SEQ_POSITIONS go through the register-resident kernels, longer
 * ones through a window-by-window path with identical results (EM passes, getR, the scorer);
 * bamm_seed_from_pwm and bamm_em_mask keep per-wave arrays over one sequence: in LDS up to about
 * 10 000 / 16 000 positions, in a global scratch region per wave beyond (same arithmetic, slower;
 * bamm_em_mask's window lists are 32 bits wide there, and from order 7 on its counts go straight into
 * the accumulator: no limit on length or order short of BAMM_MAX_ORDER, as in the reference).     */
int  bamm_seqs_upload(bamm_ctx* ctx, const bamm_packed* p, uint64_t begin, uint64_t end,
                      bamm_seqs** out);
int  bamm_seqs_destroy(bamm_seqs* s);
int  bamm_seqs_info(const bamm_seqs* s, uint64_t* n_seqs, uint64_t* total_len, uint32_t* max_len,
                    uint64_t* hbm_bytes);

/* ------------------------------------------------------------------ EM ------------------ */
typedef struct bamm_em_params {
    uint32_t K;              /* motif order             (Motif::getK)                        */
    uint32_t W;              /* motif width             (Motif::getW)                        */
    uint32_t bg_order;       /* BackgroundModel::getOrder(); EM uses min(bg_order,K) EM.cpp:23 */
    float    q;              /* Motif::getQ()           (EM.cpp:12)                          */
    int32_t  optimize_q;     /* EM ctor arg; q re-estimated in the first 5 passes of every
                              * optimize() / iterate() call (`iteration` is local to
                              * EM::optimize, EM.cpp:75-99); hand-driven accumulate/update
                              * passes: the handle's first 5                                  */
    float    epsilon;        /* EM.h:62  (0.01)                                              */
    uint32_t max_iterations; /* EM.h:63  (1000)                                              */
    uint64_t n_seqs_global;  /* N used by optimize_q (EM.cpp:515); 0 = this handle's own N   */
    uint64_t n_seqs_bound;   /* upper bound on the sequences summed into this model over ALL
                              * ranks; sizes the unit of the integer count accumulator (2^-40
                              * up to 4M sequences, coarser beyond: the int64 sums never
                              * overflow).  0 = n_seqs_global, else this handle's own count.
                              * Ranks that all-reduce together must agree on it.  A shard also
                              * plans its kernels from it and from nothing else about its size
                              * (as the whole set would: the rows a window is multiplied through
                              * decide the last bit of r), so that the model does not depend on
                              * the number of ranks or on how lengths are spread over them.     */
} bamm_em_params;

void bamm_em_default_params(bamm_em_params* p);

/* seq_mask: NULL, or n_seqs bytes (1 = sequence takes part).  vbg holds orders 0..bg_order.
 * Mirrors EM::EM (EM.cpp:7-43); v_init is Motif::getV() flattened.                           */
int  bamm_em_create(bamm_ctx* ctx, bamm_seqs* seqs, const bamm_em_params* params,
                    const float* vbg, const float* A, const float* v_init,
                    const uint8_t* seq_mask, bamm_em** out);
int  bamm_em_destroy(bamm_em* em);

/* EM::EStep (EM.cpp:139-200): s from the current v, responsibilities, log-likelihood.        */
int  bamm_em_estep(bamm_em* em);
/* EM::MStep (EM.cpp:217-259): counts from the responsibilities of the last EStep, updateV.   */
int  bamm_em_mstep(bamm_em* em);
/* EM::optimize_q (EM.cpp:505-519) from the responsibilities of the last EStep.  May be called
 * before or after the MStep (the reference calls it after, EM.cpp:93-99): getR() and MStep() keep
 * seeing the q that EStep used.                                                                */
int  bamm_em_optimize_q(bamm_em* em);
/* n fused EStep+MStep(+optimize_q while iteration<=5) passes, no convergence test, no host
 * round trip in between (benchmark / fixed-budget mode).                          */
int  bamm_em_iterate(bamm_em* em, uint32_t n);
/* EM::optimize (EM.cpp:62-137) including the stopping rule; *iterations = passes executed.   */
int  bamm_em_optimize(bamm_em* em, uint32_t* iterations);
/* EM::mask (EM.cpp:261-503, the `--advanceEM` path; f = Global::f, Global.cpp:54): an order-0
 * pass over every window, the global cut-off at the f-quantile of the responsibilities, then EM
 * restricted to the windows at or above it until the stopping rule fires.  Like both reference
 * call sites (mainBaMM.cpp:131-137, FDR.cpp:67-70) it expects a handle that has not run an
 * E-step yet.  With optimize_q the reference re-estimates q after every sequence of the first
 * pass (EM.cpp:321); that chain is inherently serial and is not offered across ranks
 * (BAMM_ERR_UNSUPPORTED when an all-reduce is installed).  W = 1 is refused: the reference
 * reads one float past its allocation there (EM.cpp:416).  *cutoff / *listed are optional.     */
int  bamm_em_mask(bamm_em* em, float f, uint32_t* iterations, float* cutoff, uint64_t* listed);

/* multi-GPU: between the local accumulation and the model update the fused accumulator
 * [n_K (4^(K+1)*W, [y][j]) | llh | sum_r | n_seqs] is summed across ranks.  It holds SIGNED 64-BIT
 * INTEGERS (fixed point: counts in units of 2^-40, llh 2^-24, sum_r 2^-30, n_seqs 1), on the device,
 * written on the context's stream: sum it as int64 (ncclInt64 / torch.int64) -- integer sums are
 * exact and order-free, so the model does not depend on the number of ranks.  The blocks of the
 * sequence kernels add into it directly; bamm_em_update consumes it and leaves it zeroed.
 * Either drive the three phases by hand ...                                                    */
int  bamm_em_accumulate(bamm_em* em);
int  bamm_em_reduce_buffer(bamm_em* em, void** dev_ptr, uint64_t* n_words);
int  bamm_em_update(bamm_em* em);
/* let the accumulator live in memory the caller allocated (e.g. a torch.int64 tensor that is handed
 * to torch.distributed.all_reduce): n_words >= 4^(K+1)*W + 3; the caller keeps ownership.        */
int  bamm_em_set_reduce_buffer(bamm_em* em, void* dev_ptr, uint64_t n_words);
/* ... or install a callback that bamm_em_iterate/optimize/mstep invoke at that point (int64 sum
 * of n_words words at dev_ptr, enqueued on hip_stream; EM::mask's window histogram goes through
 * the same callback) ...                                                                        */
typedef int (*bamm_allreduce_fn)(void* user, void* dev_ptr, uint64_t n_words, void* hip_stream);
int  bamm_em_set_allreduce(bamm_em* em, bamm_allreduce_fn fn, void* user);

/* ... or hand the handle a communicator: every pass then ends its accumulation with one
 * ncclAllReduce(ncclInt64, ncclSum) of the accumulator over RCCL / xGMI on the context's stream (the
 * reduction the reference gets from its OpenMP reduction clause and float atomics, EM.cpp:148,240,
 * 509-513).  The communicator must belong to the handle's context; it outlives the handle.       */
int  bamm_em_set_comm(bamm_em* em, bamm_comm* comm);

/* ------------------------------------------------------------------ communicators (RCCL) -- */
/* librccl is opened at the first of these calls; BAMM_ERR_COMM carries RCCL's message.         */
#define BAMM_COMM_ID_BYTES 128
/* one process, n devices: a communicator per context (distinct devices), ncclCommInitAll.  Drive
 * each context from a host thread of its own (as FDR.cpp:37 does with its folds).              */
int  bamm_comm_init_all(bamm_ctx* const* ctxs, uint32_t n, bamm_comm** out /* [n] */);
/* one process per device: rank 0 makes an id (BAMM_COMM_ID_BYTES bytes), the launcher carries it
 * to the other ranks (MPI, a file, torch.distributed ...), every rank calls init_rank.        */
int  bamm_comm_unique_id(void* id_out, size_t cap);
int  bamm_comm_init_rank(bamm_ctx* ctx, const void* id, uint32_t rank, uint32_t world, bamm_comm** out);
/* one process, n contexts on ANY devices (the same one included): the sum is staged through pinned host memory
 * between the ranks' host threads, no RCCL.  max_words >= the largest buffer summed (4^(K+1)*W + 3; EM::mask: 2049).
 * For self-tests of the N > 1 logic on a 1-GPU box and for hosts without librccl; *rccl_version reports 0.       */
int  bamm_comm_init_local(bamm_ctx* const* ctxs, uint32_t n, uint64_t max_words, bamm_comm** out /* [n] */);
/* the same between PROCESSES of one host: the sum is staged through a POSIX shared-memory segment `name` ("/something";
 * created by whichever rank comes first, unlinked once all `world` ranks are attached).  Every rank calls it with the same
 * name, world and max_words.  Every wait is bounded (a rank that does not arrive within a minute aborts the group).  For
 * self-tests -- it lets the cross-process half of the in-kernel all-reduce (hipIpc-mapped inboxes) run between two processes
 * on a 1-GPU box, where RCCL refuses two ranks on one device -- and for hosts without librccl.                          */
int  bamm_comm_init_shm(bamm_ctx* ctx, const char* name, uint32_t rank, uint32_t world, uint64_t max_words, bamm_comm** out);
int  bamm_comm_info(const bamm_comm* c, uint32_t* rank, uint32_t* world, int* rccl_version);
/* a rank that fails calls this on its communicator (any thread): the collectives its peers are blocked in return
 * BAMM_ERR_COMM instead of waiting for it for ever (ncclCommAbort; the local kind wakes its waiters).  Safe to call
 * from several threads at once on the same handle (one of them aborts, the others return).  The handle stays valid
 * for bamm_comm_destroy only; every later collective on it, and every result read from an EM handle that was using it,
 * fails with BAMM_ERR_COMM.                                                                                       */
int  bamm_comm_abort(bamm_comm* c);
int  bamm_comm_destroy(bamm_comm* c);
/* a bare loop of `iters` all-reduces of `n_words` int64 words on the context's stream (20 untimed ones first), timed with
 * HIP events: microseconds per call.  Collective: every rank of the communicator calls it with the same arguments.  What
 * a sharded iteration pays for its one collective (bench.py reports it beside the kernel time for N > 1).           */
int  bamm_comm_time_allreduce(bamm_comm* c, uint64_t n_words, uint32_t iters, float* us_per_call);
/* `count` consecutive draws of glibc's rand() stream as srand(seed) leaves it, starting behind its first `skip` draws:
 * from the restated generator (csrc/glibc_rand.h) entered by jump-ahead (use_jump != 0) or stepped `skip` times, or --
 * use_jump < 0 -- from libc's own srand(seed) / rand() (which this then advances).  *matches_libc (may be NULL): whether the
 * process-wide check found libc's rand() to be this generator.  How the N draws of Sequence.cpp:38 and the negative
 * sampler (SeqGenerator.cpp:188-348) enter the one stream in the middle, on host threads and on the device.     */
int  bamm_rand_stream_draws(uint32_t seed, uint64_t skip, int use_jump, uint32_t count, int32_t* out, int* matches_libc);
/* How this handle's passes are summed over the ranks: 0 = not at all (no communicator, no callback), 1 = one collective
 * per pass (RCCL, the host-staged group, or the caller's callback), 2 = inside the sequence kernels over peer-mapped
 * inboxes (bamm_ctx_set_tuning "peer_allreduce" = 1 on EVERY rank's context before the handle is created; default off).
 * Mode 2: the last block of a pass's launch to finish stores the GPU's totals into every peer's inbox (fine-grained device
 * memory, mapped through the process or hipIpc*; 8-byte entries that carry the pass's sequence number, so no fence and
 * no separate flag), collects the peers' entries as they arrive and leaves the sum in the accumulator, in place: no
 * collective launch behind the pass.  Same integers, same model.  Every poll is bounded ("peer_timeout_ms", default
 * 2000): a block that waits in vain raises a device flag that turns the handle's later launches into no-ops, and the
 * next result read from the handle fails with BAMM_ERR_COMM.  Applies to handles whose pass is ONE launch of a
 * grouped-column kernel built with the tail -- the mixed-row kernel (K = 2, both strands, W = 13, 14, 16, 17 or 20: the shapes
 * of BASELINE configs 2, 3 and 5) and the uniform-row kernel at K <= 2 (single strand, k = 0 / 1, other widths) -- over one
 * length class of at most 1024 positions, on 2..8 ranks; the ranks vote, and one that cannot keeps all of them on mode 1 (`note` then says why).  Collective on first use: every rank calls it (or starts its first pass). */
int  bamm_em_comm_mode(bamm_em* em, int* mode, char* note, size_t note_cap);
/* HIP devices visible to the process (0 and BAMM_ERR_NO_DEVICE when there is none)                               */
int  bamm_device_count(int* n);
/* Where a device sits and whom it reaches -- what a first multi-GPU run wants on record beside its numbers: the PCI bus id
 * ("0000:c1:00.0") of a visible device, and whether `device` can address `peer`'s memory directly (hipDeviceCanAccessPeer:
 * xGMI or PCIe peer-to-peer; the in-kernel all-reduce of bamm_em_comm_mode 2 and RCCL's direct rings need it).  No context is
 * created and nothing is enabled.                                                                                   */
int  bamm_device_pci_bus_id(int device, char* buf, size_t cap);
int  bamm_device_can_access_peer(int device, int peer, int* can);

/* results (each synchronises the stream)                                                     */
int  bamm_em_get_v(bamm_em* em, float* v_flat);        /* Motif::getV()                       */
int  bamm_em_get_counts(bamm_em* em, float* n_flat);   /* EM::n_   (EM.h:54)                  */
int  bamm_em_get_s(bamm_em* em, float* s);             /* Motif::getS() after EStep           */
int  bamm_em_get_q(bamm_em* em, float* q);             /* EM::getQ()                          */
int  bamm_em_get_llh(bamm_em* em, float* llh);         /* EM::llikelihood_ (EM.h:61)          */
int  bamm_em_get_vdiff(bamm_em* em, float* v_diff);    /* EM.cpp:102-108, last pass           */
int  bamm_em_get_iteration(bamm_em* em, uint32_t* it);
/* EM::getR() for sequences [begin,end): recomputed on demand from the s of the last EStep,
 * written in the reference's layout; out_off[n] = offset of sequence begin+n in `out`.       */
int  bamm_em_get_r(bamm_em* em, uint64_t begin, uint64_t end, float* out, uint64_t out_cap);
/* per-iteration trace of optimize()/iterate(): what the reference prints with --verbose
 * (EM.cpp:112-115).  Returns up to cap entries, *n = entries available.                      */
int  bamm_em_get_trace(bamm_em* em, float* llh, float* v_diff, float* q, uint32_t cap, uint32_t* n);
/* device time of the sequence kernel over the last iterate()/optimize() call (HIP events on
 * the context's stream): total milliseconds over the timed passes and their number.  Passes
 * 0, every, 2*every, ... of a call are timed (default every = 8; 1 = all, 0 = none): an event
 * pair takes 7-8 us of stream time, which a short iteration notices.  BAMM_TIMING_WHOLE_CALL:
 * one pair around ALL passes of the call (first event in front of the first pass's kernel, second
 * behind the last pass's): every pass covered at no cost per pass; the interval includes the
 * launch gaps and whatever runs between two passes (a collective), `launches` = the passes in it. */
#define BAMM_TIMING_WHOLE_CALL 0xffffffffu
int  bamm_em_kernel_time(bamm_em* em, float* total_ms, uint32_t* launches);
int  bamm_em_set_kernel_timing(bamm_em* em, uint32_t every);
/* how one pass is laid out: sequences that go through the grouped-column kernel (grouped.hip)
 * and through the one-column-at-a-time kernel (kernels.hip), and the kernel launches per pass. */
int  bamm_em_plan(bamm_em* em, uint64_t* grouped_seqs, uint64_t* percolumn_seqs, uint32_t* launches);
/* of the grouped ones: sequences that go through the mixed-row flavour (csrc/mixed_kernel.h: K = 2, the motif's
 * last W mod 3 groups of four columns on 6-mer rows)                                              */
int  bamm_em_plan_mixed(bamm_em* em, uint64_t* mixed_seqs);

/* ------------------------------------------------------------------ seeding ------------- */
/* The pass over the sequences of Motif::initFromPWM (Motif.cpp:228-311): 0th-order posterior of
 * every window, one motif start sampled per sequence, integer k-mer counts of the sampled sites
 * for the orders 0..K.  score = floored PWM / 0th-order background, [4][W] (Motif.cpp:205-226).
 * u[n]: the uniform variate the reference's std::discrete_distribution would draw for sequence n
 * (std::generate_canonical<double,53> on the default-seeded std::mt19937 of Motif.cpp:237, drawn
 * in sequence order for the sequences with L >= W only).  counts: v_size(K,W) ints, flat
 * [k][y][j].  z (may be NULL): the sampled index per sequence, 0 = no motif, i = window i-1.   */
int  bamm_seed_from_pwm(bamm_ctx* ctx, bamm_seqs* seqs, uint32_t K, uint32_t W, const float* score,
                        float q, const double* u, int32_t* counts, uint32_t* z);

/* ------------------------------------------------------------------ scorer -------------- */
/* ScoreSeqSet::calcLogOdds (ScoreSeqSet.cpp:25-67) with Motif::calculateLogS
 * (Motif.cpp:471-483).  mops (may be NULL): concatenated L-W+1 scores per sequence;
 * zoops[n_seqs], z[n_seqs] (first arg-max).                                                  */
int  bamm_logodds(bamm_ctx* ctx, bamm_seqs* seqs, uint32_t K, uint32_t W, uint32_t bg_order,
                  const float* v_flat, const float* vbg, float* mops, uint64_t mops_cap,
                  float* zoops, uint64_t* z);

/* same, restricted to the sequences with seq_mask[n] != 0 (the others report 0): the test /
 * negative subsets of one CV fold over a shared resident set (FDR.cpp:49-60,84-89).          */
int  bamm_logodds_subset(bamm_ctx* ctx, bamm_seqs* seqs, const uint8_t* seq_mask, uint32_t K, uint32_t W,
                         uint32_t bg_order, const float* v_flat, const float* vbg, float* mops,
                         uint64_t mops_cap, float* zoops, uint64_t* z);

/* Sequence::Sequence where the data will live: bamm_pack_codes_seeded + bamm_seqs_upload with the packing done on the
 * device (csrc/prep.hip) -- reverse complement, 2-bit stream, kmer_[i] next to unknown bases term by term with the
 * reference's rand() draws, the exception list.  The draws themselves are taken on the host (libc's one stream, entered
 * by jump-ahead on all threads).  *packed_out receives the same packed set bamm_pack_codes_seeded returns, array for
 * array (bamm_packed_free releases it); *seqs_out (may be NULL) the resident set.                                 */
int  bamm_seqs_from_codes(bamm_ctx* ctx, const uint8_t* codes, const uint64_t* off, uint64_t n_seqs, int single_strand,
                          uint32_t seed, bamm_packed** packed_out, bamm_seqs** seqs_out);
/* bamm_bg_model over a resident set: the counting pass of BackgroundModel (BackgroundModel.cpp:26-42) on the device,
 * calculateV (:441-473) on the handful of counts it leaves.  Same vbg_out as bamm_bg_model on the packed set.      */
int  bamm_seqs_bg_model(bamm_ctx* ctx, bamm_seqs* seqs, uint32_t K, const float* alpha, float* vbg_out);

/* SeqGenerator::sample_bgseqset_by_fold (SeqGenerator.cpp:63-348) on the device (csrc/negs.hip): m_fold negatives per
 * resident positive, each as long as its positive, every base one draw of the rand() stream srand(42) starts -- the
 * reference's negatives, base for base.  keep_stride > 1: only the negatives 0, stride, 2 stride, ... (with idx + stride <=
 * total) are generated and returned, which is all --FDR scores (FDR.cpp:58-60); the others still own their draws.
 * generic != 0: --genericNeg (the set's own conditionals for every positive).  *packed_out: the negatives as a packed set
 * (single strand, no exceptions; bamm_packed_free), *seqs_out (may be NULL): resident.  BAMM_ERR_UNSUPPORTED (the
 * caller samples on the host instead): s_order != 2, sequences beyond BAMM_MAX_SEQ_POSITIONS, a libc whose rand() is
 * not glibc's generator.                                                                                          */
int  bamm_sample_negatives(bamm_ctx* ctx, bamm_seqs* positives, uint32_t s_order, uint64_t m_fold, int generic,
                           uint64_t keep_stride, bamm_packed** packed_out, bamm_seqs** seqs_out);

/* ------------------------------------------------------------------ small host helpers -- */
/* BackgroundModel ctor + calculateV (BackgroundModel.cpp:3-46, :441-473): interpolated
 * order-K conditionals learned from a packed set; alpha[K+1]; vbg_out[bamm_bg_size(K)].     */
int  bamm_bg_model(const bamm_packed* p, uint32_t K, const float* alpha, float* vbg_out);
/* Motif::calculateP (Motif.cpp:430-469); host arithmetic on <= 41k elements.                 */
int  bamm_calculate_p(const float* v_flat, const float* vbg, uint32_t bg_order, uint32_t K,
                      uint32_t W, float* p_flat);
size_t bamm_v_size(uint32_t K, uint32_t W);
size_t bamm_v_offset(uint32_t k, uint32_t W);
size_t bamm_bg_size(uint32_t K);

#ifdef __cplusplus
}
#endif
#endif /* BAMM_EM_H_ */
