// TEST INFRASTRUCTURE ONLY -- never linked or loaded by the product path.
//
// Driver over the *real* reference classes (compiled in place from
// /root/reference/src by oracle/Makefile into oracle/_ref/libbammref.so).
// It exposes a flat extern "C" surface so that tests / the golden-vector
// generator can run the reference's own EM::EStep/MStep/optimize,
// BackgroundModel, Motif::updateV/calculateP/write and
// ScoreSeqSet::calcLogOdds on arbitrary encoded sequences.
//
// Only reference translation units that build with the stock toolchain are
// used: init/{Alphabet,Sequence,BackgroundModel,Motif}.cpp, refinement/EM.cpp,
// seq_scoring/ScoreSeqSet.cpp.  init/SequenceSet.cpp (FASTA reader) needs
// Boost, which this image lacks; it is treated as unbuildable and NOT stubbed.
// Consequently Motif::initFromPWM (takes a SequenceSet*) is never called here.
//
// This file contains no reference code: it only calls public members
// (EM.h:20-36, Motif.h:14-45, BackgroundModel.h:22-57, ScoreSeqSet.h:26-36)
// plus two private EM fields (llikelihood_, n_) read through the usual
// `#define private public` test trick, applied to the reference headers only.

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <iomanip>
#include <limits>
#include <numeric>
#include <string>
#include <vector>
#include <algorithm>
#include <memory>
#include <utility>
#include <random>
#include <chrono>
#include <assert.h>
#include <math.h>
#include <float.h>
#include <sys/stat.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define private public
#include "refinement/EM.h"
#include "seq_scoring/ScoreSeqSet.h"
#include "seq_generator/SeqGenerator.h"
#include "evaluation/FDR.h"
#undef private

namespace {

struct Session {
    std::vector<Sequence*> seqs;
};

size_t flat_size(size_t K, size_t W) {
    size_t n = 0, y = 4;
    for (size_t k = 0; k <= K; k++) { n += y * W; y *= 4; }
    return n;
}

void flatten(float*** t, size_t K, size_t W, float* out) {
    size_t Y = 4, o = 0;
    for (size_t k = 0; k <= K; k++) {
        for (size_t y = 0; y < Y; y++)
            for (size_t j = 0; j < W; j++) out[o++] = t[k][y][j];
        Y *= 4;
    }
}

bool g_alphabet_ready = false;

}  // namespace

extern "C" {

void ref_set_threads(int n) {
#ifdef _OPENMP
    omp_set_num_threads(n);
#else
    (void)n;
#endif
}

// codes: Alphabet encoding (0 = N, 1..4 = A,C,G,T), concatenated; offsets[N+1].
// Mirrors mainBaMM.cpp:22 (srand) + SequenceSet.cpp's per-record Sequence ctor call.
void* ref_session_create(const uint8_t* codes, const uint64_t* offsets, uint64_t N,
                         int single_strand, int do_srand, unsigned seed) {
    if (!g_alphabet_ready) {
        char name[] = "STANDARD";
        Alphabet::init(name);
        g_alphabet_ready = true;
    }
    if (do_srand) srand(seed);
    Session* s = new Session();
    std::vector<size_t> Y;  // ignored by the Sequence ctor
    for (uint64_t n = 0; n < N; n++) {
        size_t L = offsets[n + 1] - offsets[n];
        std::vector<uint8_t> tmp(codes + offsets[n], codes + offsets[n + 1]);
        s->seqs.push_back(new Sequence(tmp.data(), L, "seq" + std::to_string(n), Y, single_strand != 0));
    }
    return s;
}

void ref_session_destroy(void* h) {
    Session* s = static_cast<Session*>(h);
    for (auto* q : s->seqs) delete q;
    delete s;
}

uint64_t ref_seq_L(void* h, uint64_t n) { return static_cast<Session*>(h)->seqs[n]->getL(); }
const uint64_t* ref_seq_kmer(void* h, uint64_t n) {
    static_assert(sizeof(size_t) == sizeof(uint64_t), "size_t");
    return reinterpret_cast<const uint64_t*>(static_cast<Session*>(h)->seqs[n]->getKmer());
}
const uint8_t* ref_seq_codes(void* h, uint64_t n) { return static_cast<Session*>(h)->seqs[n]->getSequence(); }

// ---- background model (BackgroundModel.cpp:3-46, :441-473) ----
void* ref_bg_create(void* h, uint64_t order, const float* alpha) {
    Session* s = static_cast<Session*>(h);
    std::vector<float> A(alpha, alpha + order + 1);
    return new BackgroundModel(s->seqs, order, A, true, "ref");
}
void ref_bg_destroy(void* b) { delete static_cast<BackgroundModel*>(b); }
const float* ref_bg_v(void* b, uint64_t k) { return static_cast<BackgroundModel*>(b)->getV()[k]; }
void ref_bg_write(void* b, const char* dir, const char* base) {
    std::string d(dir);
    static_cast<BackgroundModel*>(b)->write(&d[0], base);
}

// ---- motif (Motif.cpp:5-54 ctor, :56-101 copy ctor sets isInitialized_) ----
void* ref_motif_create(uint64_t W, uint64_t K, const float* alpha, void* bg, float q, const float* v_flat) {
    BackgroundModel* b = static_cast<BackgroundModel*>(bg);
    std::vector<float> A(alpha, alpha + K + 1);
    Motif blank(W, K, A, b->getV(), b->getOrder(), q);
    float*** v = blank.getV();
    size_t Y = 4, o = 0;
    for (size_t k = 0; k <= K; k++) {
        for (size_t y = 0; y < Y; y++)
            for (size_t j = 0; j < W; j++) v[k][y][j] = v_flat[o++];
        Y *= 4;
    }
    return new Motif(blank);  // copy ctor marks the motif initialised
}
void* ref_motif_from_bamm_file(uint64_t W, uint64_t K, const float* alpha, void* bg, float q, const char* path) {
    BackgroundModel* b = static_cast<BackgroundModel*>(bg);
    std::vector<float> A(alpha, alpha + K + 1);
    Motif* m = new Motif(W, K, A, b->getV(), b->getOrder(), q);
    std::string p(path);
    m->initFromBaMM(&p[0], 0, 0);
    return m;
}
void ref_motif_destroy(void* m) { delete static_cast<Motif*>(m); }
uint64_t ref_motif_flat_size(void* m) {
    Motif* mo = static_cast<Motif*>(m);
    return flat_size(mo->getK(), mo->getW());
}
void ref_motif_get_v(void* m, float* out) {
    Motif* mo = static_cast<Motif*>(m);
    flatten(mo->getV(), mo->getK(), mo->getW(), out);
}
void ref_motif_get_p(void* m, float* out) {  // after calculateP (Motif.cpp:430-469)
    Motif* mo = static_cast<Motif*>(m);
    mo->calculateP();
    flatten(mo->p_, mo->getK(), mo->getW(), out);
}
void ref_motif_get_s(void* m, float* out) {  // s_[y][j], Y[K+1] x W
    Motif* mo = static_cast<Motif*>(m);
    size_t Y = 1;
    for (size_t k = 0; k <= mo->getK(); k++) Y *= 4;
    float** s = mo->getS();
    for (size_t y = 0; y < Y; y++)
        for (size_t j = 0; j < mo->getW(); j++) out[y * mo->getW() + j] = s[y][j];
}
void ref_motif_linear_s(void* m, void* bg, uint64_t K_bg) {
    static_cast<Motif*>(m)->calculateLinearS(static_cast<BackgroundModel*>(bg)->getV(), K_bg);
}
void ref_motif_log_s(void* m, void* bg, uint64_t K_bg) {
    static_cast<Motif*>(m)->calculateLogS(static_cast<BackgroundModel*>(bg)->getV(), K_bg);
}
void ref_motif_write(void* m, const char* dir, const char* base) {
    std::string d(dir);
    static_cast<Motif*>(m)->write(&d[0], base);
}

// ---- EM (EM.cpp:7-43 ctor, :62-137 optimize, :139-200 EStep, :217-259 MStep, :505-519 optimize_q) ----
void* ref_em_create(void* m, void* bg, void* h, int optimizeQ, int verbose, float f) {
    Session* s = static_cast<Session*>(h);
    return new EM(static_cast<Motif*>(m), static_cast<BackgroundModel*>(bg), s->seqs, optimizeQ != 0, verbose != 0, f);
}
void ref_em_destroy(void* e) { delete static_cast<EM*>(e); }
void ref_em_estep(void* e) { static_cast<EM*>(e)->EStep(); }
void ref_em_mstep(void* e) { static_cast<EM*>(e)->MStep(); }
void ref_em_optimize_q(void* e) { static_cast<EM*>(e)->optimize_q(); }
int ref_em_optimize(void* e) { return static_cast<EM*>(e)->optimize(); }
int ref_em_mask(void* e) { return static_cast<EM*>(e)->mask(); }          // EM.cpp:261-503 (--advanceEM)
float ref_em_q(void* e) { return static_cast<EM*>(e)->getQ(); }
float ref_em_llh(void* e) { return static_cast<EM*>(e)->llikelihood_; }
const float* ref_em_r(void* e, uint64_t n) { return static_cast<EM*>(e)->getR()[n]; }
void ref_em_get_n(void* e, float* out) {
    EM* em = static_cast<EM*>(e);
    flatten(em->n_, em->K_, em->W_, out);
}
void ref_em_write(void* e, const char* dir, const char* base, int ss) {
    std::string d(dir);
    static_cast<EM*>(e)->write(&d[0], base, ss != 0);
}

// ---- scorer (ScoreSeqSet.cpp:25-67) ----
// mops_out: concatenated LW1 scores per sequence; zoops_out[N]; z_out[N]
void ref_logodds(void* m, void* bg, void* h, float* mops_out, float* zoops_out, uint64_t* z_out) {
    Session* s = static_cast<Session*>(h);
    ScoreSeqSet sc(static_cast<Motif*>(m), static_cast<BackgroundModel*>(bg), s->seqs);
    sc.calcLogOdds();
    std::vector<std::vector<float>> mops = sc.getMopsScores();
    std::vector<float> zoops = sc.getZoopsScores();
    size_t o = 0;
    for (size_t n = 0; n < mops.size(); n++) {
        for (float x : mops[n]) mops_out[o++] = x;
        zoops_out[n] = zoops[n];
        z_out[n] = sc.z_[n];
    }
}

// ScoreSeqSet::writeLogOdds (ScoreSeqSet.cpp:293-331): <dir>/<base>.logOddsZoops
void ref_write_logodds(void* m, void* bg, void* h, const char* dir, const char* base, int ss) {
    Session* s = static_cast<Session*>(h);
    ScoreSeqSet sc(static_cast<Motif*>(m), static_cast<BackgroundModel*>(bg), s->seqs);
    sc.calcLogOdds();
    std::string d(dir);
    sc.writeLogOdds(&d[0], base, ss != 0);
}

// ---- negative set (SeqGenerator.cpp:3-42 ctor incl. srand(42), :188-204 sample_bgseqset_by_fold) ----
void* ref_negset_create(void* h, uint64_t sOrder, uint64_t mFold, int genericNeg) {
    Session* s = static_cast<Session*>(h);
    SeqGenerator gen(s->seqs, NULL, sOrder, 1.0f, genericNeg != 0);
    std::vector<std::unique_ptr<Sequence>> neg = gen.sample_bgseqset_by_fold(mFold);
    Session* out = new Session();
    for (auto& q : neg) out->seqs.push_back(q.release());
    return out;
}
uint64_t ref_session_size(void* h) { return static_cast<Session*>(h)->seqs.size(); }

// ---- ScoreSeqSet::calcPvalues + write (.occurrence)  (ScoreSeqSet.cpp:70-126, :245-291) ----
// neg_all: all MOPS scores of the negative set (mainBaMM.cpp:215-221)
void ref_occurrence(void* m, void* bg, void* h, const float* neg_all, uint64_t n_neg, float pval_cutoff, int ss,
                    const char* dir, const char* base, float* pvalues_out) {
    Session* s = static_cast<Session*>(h);
    ScoreSeqSet sc(static_cast<Motif*>(m), static_cast<BackgroundModel*>(bg), s->seqs);
    sc.calcLogOdds();
    std::vector<float> neg(neg_all, neg_all + n_neg);
    sc.calcPvalues(sc.getMopsScores(), neg);
    std::string d(dir);
    sc.write(&d[0], base, pval_cutoff, ss != 0);
    size_t o = 0;
    for (auto& v : sc.mops_p_values_) for (float x : v) pvalues_out[o++] = x;
}

// ---- FDR (FDR.cpp:3-27 ctor, :28-145 evaluateMotif, :147-276 calculatePR, :278-333 calculatePvalues, :338-450 write) ----
void* ref_fdr_create(void* pos, void* neg, void* m, void* bg, uint64_t cvFold, int mops, int zoops, int savePRs,
                     int savePvalues, int saveLogOdds) {
    return new FDR(static_cast<Session*>(pos)->seqs, static_cast<Session*>(neg)->seqs, static_cast<Motif*>(m),
                   static_cast<BackgroundModel*>(bg), cvFold, mops != 0, zoops != 0, savePRs != 0, savePvalues != 0,
                   saveLogOdds != 0);
}
void ref_fdr_destroy(void* f) { delete static_cast<FDR*>(f); }
void ref_fdr_evaluate(void* f, int em, int optimizeQ, float frac, uint64_t threads) {
    static_cast<FDR*>(f)->evaluateMotif(em != 0, false, optimizeQ != 0, false, frac, threads);
}
void ref_fdr_write(void* f, const char* dir, const char* base) {
    std::string d(dir);
    static_cast<FDR*>(f)->write(&d[0], base);
}
// which: 0 posScoreMax_, 1 negScoreMax_, 2 posScoreAll_, 3 negScoreAll_ (as left by calculatePR/Pvalues: sorted)
uint64_t ref_fdr_scores(void* f, int which, float* out, uint64_t cap) {
    FDR* fd = static_cast<FDR*>(f);
    std::vector<float>& v = which == 0 ? fd->posScoreMax_ : which == 1 ? fd->negScoreMax_ : which == 2 ? fd->posScoreAll_ : fd->negScoreAll_;
    for (size_t i = 0; i < v.size() && i < cap; i++) out[i] = v[i];
    return v.size();
}
float ref_fdr_q(void* f) { return static_cast<FDR*>(f)->q_; }
// run only the statistics on caller-provided scores (unit-test entry for the host restatement)
void ref_fdr_stats_only(void* f, const float* pos_max, uint64_t n_pos_max, const float* neg_max, uint64_t n_neg_max,
                        const float* pos_all, uint64_t n_pos_all, const float* neg_all, uint64_t n_neg_all,
                        int with_pvalues) {
    FDR* fd = static_cast<FDR*>(f);
    fd->posScoreMax_.assign(pos_max, pos_max + n_pos_max);
    fd->negScoreMax_.assign(neg_max, neg_max + n_neg_max);
    fd->posScoreAll_.assign(pos_all, pos_all + n_pos_all);
    fd->negScoreAll_.assign(neg_all, neg_all + n_neg_all);
    fd->calculatePR();
    if (with_pvalues) fd->calculatePvalues();
}

}  // extern "C"
