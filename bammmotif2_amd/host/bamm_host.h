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
struct FastaSet {
    std::vector<std::string> headers;   // keep the leading '>' like the reference (SequenceSet.cpp:137-139)
    std::vector<uint8_t> codes;         // 0 = N, 1..4 = A,C,G,T
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
// Motif::initFromPWM (Motif.cpp:192-333): pwm[y][j] (4 x W); yK = kmer_ mod 4^(K+1) per position
void motif_init_from_pwm(Motif& m, const std::vector<float>& pwm, const BgModel& bg, const uint32_t* yK,
                         const uint64_t* off, size_t n_seqs, float q);
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
               const uint64_t* off, size_t n_seqs, SeedSet& out, std::string& err);

std::string base_name(const std::string& path);   // refinement/utils.h:66-85

}  // namespace bammhost
