/* TEST INFRASTRUCTURE ONLY -- CPU restatement of the BaMMmotif2 EM hot path.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library, and only as the checker.  The product path (bammmotif2_amd/) never does.
 *
 * Parity status: PINNED.  Every function below is checked bit-for-bit (single thread) or
 * to <= 2e-7 (multi-thread) against the real reference classes compiled in place into
 * oracle/_ref/libbammref.so (tests/test_oracle_vs_reference.py, runs in the dev container)
 * and against the committed outputs of that build in tests/golden/ (runs everywhere).
 * Exceptions: orc_init_from_pwm and the FASTA reader have no runnable reference here
 * (init/SequenceSet.cpp needs Boost) -- they are pinned only by the probe numbers recorded
 * in SURVEY.md section 6 (JunD: 13 iterations, llh 454.708 -> 475.799) and are marked
 * "parity unpinned" in DESIGN.md.
 *
 * All file:line citations are relative to /root/reference/src.
 *
 * Flat layouts
 *   kmer   : uint64 per position, sequences concatenated; off[N+1] gives the start of each
 *            sequence, L_n = off[n+1]-off[n]                        (Sequence.h:47 kmer_)
 *   v, n, p: [k][y][j] row-major, orders concatenated; order k starts at W*(4^(k+1)-4)/3
 *                                                                  (Motif.h:57-63)
 *   vbg    : [k][y], order k starts at (4^(k+1)-4)/3               (BackgroundModel.h:66)
 *   s      : [y][j], 4^(K+1) x W                                   (Motif.h:62)
 *   A      : [k][j], (K+1) x W                                     (Motif.h:56, Motif.cpp:43-46)
 *   r      : float per position, same offsets as kmer, reference's reversed index:
 *            r[off[n] + L-W-i] is the responsibility of window start i (EM.cpp:173)
 */
#ifndef BAMM_ORACLE_H_
#define BAMM_ORACLE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

void   orc_set_threads(int n);
size_t orc_v_size(size_t K, size_t W);   /* floats in a flat v / n / p tensor */
size_t orc_v_offset(size_t k, size_t W); /* start of order k                 */
size_t orc_bg_size(size_t K);
size_t orc_bg_offset(size_t k);

/* Sequence.cpp:4-43, :91-99 (+ Alphabet.cpp:46-55).  codes: 0=N, 1..4=ACGT.
 * Writes seq_out[L] (L = 2*L0+1, or L0 when single_strand) and kmer_out[L]; consumes libc
 * rand() exactly like the reference (one draw per (position, digit) term whose base is 0). */
size_t orc_seq_length(size_t L0, int single_strand);
void   orc_encode_sequence(const uint8_t* codes, size_t L0, int single_strand,
                           uint8_t* seq_out, uint64_t* kmer_out);
/* whole set, in file order, optionally after srand(seed) (mainBaMM.cpp:22) */
void   orc_encode_set(const uint8_t* codes, const uint64_t* in_off, size_t N, int single_strand,
                      int do_srand, unsigned seed, uint8_t* seq_out, uint64_t* kmer_out,
                      uint64_t* out_off);

/* BackgroundModel.cpp:3-46 (counts) + :441-473 (calculateV, interpolated) */
void   orc_bg_model(const uint64_t* kmer, const uint64_t* off, size_t N, size_t K,
                    const float* alpha, float* vbg_out);

/* Motif.cpp:485-494 / :471-483 */
void   orc_linear_s(const float* v, const float* vbg, size_t K, size_t W, size_t K_bg, float* s);
void   orc_log_s(const float* v, const float* vbg, size_t K, size_t W, size_t K_bg, float* s);

/* EM.cpp:139-200.  r must hold off[N] floats; fully overwritten.  Returns llikelihood_. */
float  orc_estep(const uint64_t* kmer, const uint64_t* off, size_t N, size_t K, size_t W,
                 const float* s, float q, float* r);

/* EM.cpp:217-254: zero n, accumulate order K (CAS float atomics under OpenMP), marginalise. */
void   orc_mstep_counts(const uint64_t* kmer, const uint64_t* off, size_t N, size_t K, size_t W,
                        const float* r, float* n);

/* Motif.h:95-136 */
void   orc_update_v(const float* n, const float* A, const float* vbg, size_t K, size_t W, float* v);

/* EM.cpp:505-519 */
float  orc_optimize_q(const float* r, const uint64_t* off, size_t N, size_t W);

/* Motif.cpp:430-469 */
void   orc_calculate_p(const float* v, const float* vbg, size_t k_bg, size_t K, size_t W, float* p);

/* EM.cpp:62-137.  v is updated in place, q_io updated when optimizeQ.  trace_llh / trace_vdiff
 * (each max_iter floats, may be NULL) receive the per-iteration values the reference prints
 * in verbose mode.  r (off[N] floats) and n (orc_v_size floats) hold the last E/M results.
 * epsilon and max_iter are the reference's hard-coded 0.01 / 1000 (EM.h:62-63) unless the
 * caller overrides them.  Returns the number of iterations executed. */
size_t orc_optimize(const uint64_t* kmer, const uint64_t* off, size_t N, size_t K, size_t W,
                    size_t bg_order, const float* vbg, const float* A, float* v, float* q_io,
                    int optimizeQ, float epsilon, size_t max_iter, float* r, float* n,
                    float* trace_llh, float* trace_vdiff, float* llh_out);

/* EM.cpp:261-503 (--advanceEM): order-0 pass over all windows (own boundary rule, :295-305, and
 * optimize_q() inside the sequence loop, :321), global cut-off at the f-quantile of r by a full
 * descending sort (:329-343), then EM restricted to the listed windows (full-width products, no
 * truncation, serial M-step), with the reference's indexing as written (:416 pos_[n][LW1-ri],
 * :421 r_[n][0] /= normFactor).  r (off[N] floats) is treated as freshly calloc'ed, as at both
 * reference call sites (mainBaMM.cpp:131-137, FDR.cpp:67-70).  Returns the iteration count. */
size_t orc_mask(const uint64_t* kmer, const uint64_t* off, size_t N, size_t K, size_t W,
                size_t bg_order, const float* vbg, const float* A, float* v, float* q_io,
                int optimizeQ, float f, float epsilon, size_t max_iter, float* r, float* n,
                float* trace_llh, float* trace_vdiff, float* llh_out, float* cutoff_out,
                uint64_t* listed_out);

/* ScoreSeqSet.cpp:25-67.  mops: concatenated LW1 scores per sequence (mops_off[n] =
 * sum_{m<n} (L_m-W+1)); zoops[N]; z[N] (first arg-max). */
void   orc_logodds(const uint64_t* kmer, const uint64_t* off, size_t N, size_t K, size_t W,
                   const float* s_log, float* mops, float* zoops, uint64_t* z);

/* Motif.cpp:192-333 (serial; default-constructed std::mt19937 + libstdc++'s
 * std::discrete_distribution restated).  pwm: [y][j] 4 x W.  v_out: flat v. */
void   orc_init_from_pwm(const float* pwm, size_t W, size_t K, const float* A, const float* vbg,
                         const uint64_t* kmer, const uint64_t* off, size_t N, float q, float* v_out);
/* the same, also reporting the sampled site per sequence (z_out[n]: 0 = no motif, i = window i-1) and the
 * integer site counts of all orders, flat [k][y][j]; either may be NULL */
void   orc_init_from_pwm_sites(const float* pwm, size_t W, size_t K, const float* A, const float* vbg,
                               const uint64_t* kmer, const uint64_t* off, size_t N, float q, float* v,
                               uint32_t* z_out, int* counts_out);

/* fp64 restatement of one E+M step (precision oracle): same formulas,
 * every product / sum in double; outputs rounded to float at the end. */
void   orc_em_step_f64(const uint64_t* kmer, const uint64_t* off, size_t N, size_t K, size_t W,
                       size_t bg_order, const float* vbg, const float* A, const float* v_in,
                       float q, float* v_out, float* n_out, double* llh_out, double* sum_r_out);

#ifdef __cplusplus
}
#endif
#endif
