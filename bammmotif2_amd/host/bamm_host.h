// C++17 host side of the drop-in `BaMMmotif OUTDIR FASTA [--EM ...]` path: FASTA reader, model
// file I/O (.hbcp/.hbp/.ihbcp/.ihbp), seeds (MEME PWM / BaMM file / binding sites) and the EM
// driver over the C ABI (include/bamm_em.h).  Behaviour follows the reference files cited at
// each function (paths relative to /root/reference/src); the code is written from scratch.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/bamm_em.h"

namespace bammhost {

// ---- sequences (init/SequenceSet.cpp:67-225, init/Alphabet.cpp:10-55, STANDARD alphabet) ----
// resize() of a vector of bytes with this allocator leaves the new bytes uninitialised: 400 MB of negatives are written
// by the sampler's threads, a serial zero-fill in front of that costs as much as the sampling (1.0 of 1.6 s)
template <class T>
struct DefaultInitAlloc : std::allocator<T> {
    template <class U> struct rebind { using other = DefaultInitAlloc<U>; };
    template <class U> void construct(U* p) noexcept { ::new (static_cast<void*>(p)) U; }
    template <class U, class... A> void construct(U* p, A&&... a) { ::new (static_cast<void*>(p)) U(std::forward<A>(a)...); }
};
using ByteVec = std::vector<uint8_t, DefaultInitAlloc<uint8_t>>;

struct FastaSet {
    std::vector<std::string> headers;   // keep the leading '>' like the reference (SequenceSet.cpp:137-139)
    ByteVec codes;                      // 0 = N, 1..4 = A,C,G,T
    std::vector<uint64_t> off;          // [n+1]
    size_t min_len = 0, max_len = 0;
    float base_freq[4] = {0, 0, 0, 0};
    size_t size() const { return headers.size(); }
};
// returns 0, or 1 with `err` holding the reference's message (caller prints + exit(1))
int read_fasta(const std::string& path, FastaSet& out, std::string& err);

// ---- background model (init/BackgroundModel.cpp) ----
struct BgModel {
    uint32_t K = 2;
    std::vector<float> alpha;           // K+1
    std::vector<float> v;               // flat [k][y]
};
int bg_learn(const bamm_packed* p, uint32_t K, const std::vector<float>& alpha, BgModel& out);   // :3-46, :441-473
int bg_read(const std::string& path, BgModel& out, std::string& err);                            // :48-129
int bg_write(const std::string& dir, const std::string& basename, const BgModel& bg, std::string& err);  // :353-439

// ---- motif (init/Motif.{h,cpp}, init/MotifSet.cpp) ----
struct Motif {
    uint32_t W = 0, K = 0;
    float q = 0.3f;
    std::vector<float> alpha;           // K+1
    std::vector<float> A;               // (K+1) x W, A[k][j] = alpha[k]  (Motif.cpp:43-46)
    std::vector<float> v, p;            // flat [k][y][j]
};
void motif_alloc(Motif& m, uint32_t W, uint32_t K, const std::vector<float>& alpha, float q);
// a resident copy of the training set: lets initFromPWM's pass over the sequences run on the device
struct SeedDevice {
    bamm_ctx* ctx = nullptr;
    bamm_seqs* seqs = nullptr;
};
// Motif::initFromPWM (Motif.cpp:192-333): pwm[y][j] (4 x W); yK = kmer_ mod 4^(K+1) per position (host
// path; may be null with `dev`).  With `dev` the posteriors, the sampling and the counts run through
// bamm_seed_from_pwm; the std::mt19937 draws stay here, in sequence order.  Returns 0, or 1 + err.
// z_out (optional): the sampled site per sequence, 0 = no motif, i = window i-1
int motif_init_from_pwm(Motif& m, const std::vector<float>& pwm, const BgModel& bg, const uint32_t* yK,
                        const uint64_t* off, size_t n_seqs, float q, const SeedDevice* dev, std::string& err,
                        std::vector<uint32_t>* z_out = nullptr);
int motif_init_from_bamm(Motif& m, const std::string& path, uint32_t l_flank, uint32_t r_flank, const BgModel& bg,
                         std::string& err);                                                     // Motif.cpp:336-397
int motif_init_from_sites(Motif& m, const std::string& path, uint32_t l_flank, uint32_t r_flank, const BgModel& bg,
                          std::string& err);                                                    // Motif.cpp:134-189
void motif_calculate_p(Motif& m, const BgModel& bg);                                             // Motif.cpp:430-469
int motif_write(const std::string& dir, const std::string& basename, const Motif& m, std::string& err);  // Motif.cpp:515-547

struct SeedSet {
    std::vector<Motif> motifs;
    uint32_t max_w = 0;
};
// MotifSet::MotifSet (MotifSet.cpp:3-222); tag = "PWM" | "BaMM" | "bindingsites"
int load_seeds(const std::string& path, const std::string& tag, uint32_t l_flank, uint32_t r_flank, uint32_t K,
               const std::vector<float>& alpha, size_t max_pwm, float glob_q, const BgModel& bg, const uint32_t* yK,
               const uint64_t* off, size_t n_seqs, SeedSet& out, std::string& err, const SeedDevice* dev = nullptr);

std::string base_name(const std::string& path);   // refinement/utils.h:66-85

}  // namespace bammhost

// ======================= evaluation side: negatives, FDR statistics, occurrences ===================
namespace bammhost {

// SeqGenerator::sample_bgseqset_by_fold (seq_generator/SeqGenerator.cpp:188-204) incl. the ctor's
// srand(42) (:35), calculate_kmer_frequency (:63-110), rescale_kmer_frequency (:112-186, hard-wired
// to s = 2 like the reference), bgseq_on_rescaled_v (:285-348) and bg_sequence (:222-283).
// y_s: kmer_ mod 4^(s+1) for every position of the reference sequences (as EM sees them).

// every core the process may use (affinity mask capped by the cgroup quota): for host work whose result does not
// depend on how it is cut -- the negative sampler, packing, sorts
int host_parallelism();
void set_host_parallelism(int n);           // tests: force a thread count (0 = every granted core again)
int sample_negatives(const uint32_t* y_s, const uint64_t* off, size_t n_seqs, uint32_t s_order, size_t m_fold,
                     bool generic, ByteVec& codes_out, std::vector<uint64_t>& off_out, std::string& err, size_t keep_stride = 0);

// one float as `ostream << float` prints it at `precision` significant digits (printf %g); `out` holds 48 bytes; returns the length
size_t format_g(char* out, float x, int precision);

struct FdrResult {                      // what FDR::calculatePR / calculatePvalues leave behind (FDR.h:54-84)
    std::vector<float> zoops_tp, zoops_fp, zoops_fdr, zoops_rec, pn_pvalue, zoops_pvalue;
    std::vector<float> mops_tp, mops_fp, mops_fdr, mops_rec, mops_pvalue;
    float occ_frac = 0.f, occ_mult = 0.f;
};
// scores may arrive in any order (they are sorted first, FDR.cpp:158-159,203-204)
void fdr_statistics(std::vector<float> pos_max, std::vector<float> neg_max, std::vector<float> pos_all,
                    std::vector<float> neg_all, size_t posN, size_t negN, float q, bool mops, bool zoops,
                    bool with_pvalues, FdrResult& out);
int fdr_write(const std::string& dir, const std::string& basename, const FdrResult& r, size_t posN, size_t negN,
              bool mops, bool zoops, bool save_prs, bool save_pvalues, std::string& err);   // FDR.cpp:338-410

// --saveLogOdds: FDR::write's .zoops.logOdds / .mops.logOdds (FDR.cpp:416-450) and
// ScoreSeqSet::writeLogOdds' .logOddsZoops (ScoreSeqSet.cpp:293-331)
int fdr_logodds_write(const std::string& dir, const std::string& basename, std::vector<float> pos_max, std::vector<float> neg_max,
                      std::vector<float> pos_all, std::vector<float> neg_all, size_t posN, size_t negN, bool mops, bool zoops,
                      bool ascending, std::string& err);
int logodds_zoops_write(const std::string& dir, const std::string& basename, const std::vector<std::string>& headers,
                        const uint8_t* codes, const uint64_t* off, size_t n_seqs, bool revcomp, bool ss, uint32_t W,
                        const float* zoops, const uint64_t* z, std::string& err);

// ScoreSeqSet::calcPvalues (seq_scoring/ScoreSeqSet.cpp:70-126): p-/e-values of every window
void mops_pvalues(const float* pos_scores, size_t n_pos_scores, std::vector<float> neg_all, size_t posN,
                  std::vector<float>& p_out, std::vector<float>& e_out);

// ScoreSeqSet::write (seq_scoring/ScoreSeqSet.cpp:245-291): <basename>.occurrence, one row per window with
// p < cutoff.  codes/off: FASTA codes of the forward strands; ss as on the command line.
int occurrence_write(const std::string& dir, const std::string& basename, const std::vector<std::string>& headers,
                     const uint8_t* codes, const uint64_t* off, size_t n_seqs, bool ss, uint32_t W, const float* p,
                     const float* e, float cutoff, std::string& err);

}  // namespace bammhost
