// EM_hip.cpp -- what a BaMMmotif2 maintainer drops in place of src/refinement/EM.cpp to run the EM
// refinement on an MI355X through libbamm_em.so (include/bamm_em.h).
//
// src/refinement/EM.h is used AS IT IS (not a line changed): the device handles live in a side table keyed
// by the object, every public member of `class EM` (EM.h:20-36) is defined here, and the private members the
// rest of the reference or a debugger may look at (q_, llikelihood_, n_, r_) are kept up to date.  The
// caller's Motif is updated in place after every step, as the reference does (Motif::updateV through
// EM::MStep, EM.cpp:258).
//
// Built and exercised, not just shown: `make -C oracle ref_hip` compiles the reference's own translation
// units with this file and ScoreSeqSet_hip.cpp in place of refinement/EM.cpp and
// seq_scoring/ScoreSeqSet.cpp (oracle/_ref/libbammref_hip.so); tests/test_integration_gpu.py drives the
// reference's `EM`, `FDR` and `ScoreSeqSet` classes through it and reproduces the golden vectors that the
// unmodified reference produced.
#include "refinement/EM.h"

#include <chrono>
#include <cstring>
#include <mutex>
#include <unordered_map>

#include "bamm_em.h"

namespace {

struct Device {                                          // what EM.h has no member for
    bamm_ctx* ctx = nullptr;
    bamm_seqs* seqs = nullptr;
    bamm_em* em = nullptr;
    uint64_t total_positions = 0;
    bool r_fresh = false;                                // r_ holds the responsibilities of the last E pass
};

std::mutex g_mu;                                         // FDR::evaluateMotif builds EM objects on several threads (FDR.cpp:37)
std::unordered_map<const EM*, Device> g_dev;

Device& dev(const EM* self) {
    std::lock_guard<std::mutex> lock(g_mu);
    return g_dev[self];
}

[[noreturn]] void die(const char* what) {                // reference style: message on stderr + exit(1)
    std::cerr << "Error: " << what << ": " << bamm_last_error() << std::endl;
    exit(1);
}

}  // namespace

EM::EM(Motif* motif, BackgroundModel* bgModel, std::vector<Sequence*> seqs, bool optimizeQ, bool verbose, float f) {
    motif_ = motif;                                      // EM.cpp:7-43
    bgModel_ = bgModel;
    q_ = motif->getQ();
    f_ = f;
    seqs_ = seqs;
    optimizeQ_ = optimizeQ;
    verbose_ = verbose;
    K_ = motif_->getK();
    W_ = motif_->getW();
    Y_ = motif_->getY();
    s_ = motif_->getS();
    A_ = motif_->getA();
    K_bg_ = (bgModel_->getOrder() < K_) ? bgModel_->getOrder() : K_;
    // r_ (EM::getR) is materialised on demand, pos_ is a constant the reference only ever writes (EM.cpp:160-164)
    r_ = nullptr;
    pos_ = nullptr;
    n_ = (float***)calloc(K_ + 1, sizeof(float**));      // EM.cpp:34-40
    for (size_t k = 0; k < K_ + 1; k++) {
        n_[k] = (float**)calloc(Y_[k + 1], sizeof(float*));
        for (size_t y = 0; y < Y_[k + 1]; y++) n_[k][y] = (float*)calloc(W_, sizeof(float));
    }

    Device& d = dev(this);
    if (bamm_ctx_create(0, nullptr, &d.ctx)) die("no usable MI355X");
    std::vector<const uint64_t*> km(seqs_.size());       // Sequence::getKmer() is size_t* == uint64_t* on LP64
    std::vector<uint64_t> L(seqs_.size());
    for (size_t n = 0; n < seqs_.size(); n++) {
        km[n] = reinterpret_cast<const uint64_t*>(seqs_[n]->getKmer());
        L[n] = seqs_[n]->getL();
        d.total_positions += L[n];
    }
    bamm_packed* pk = nullptr;
    if (bamm_pack_kmer_ptrs(km.data(), L.data(), seqs_.size(), &pk)) die("packing the sequences");
    if (bamm_seqs_upload(d.ctx, pk, 0, seqs_.size(), &d.seqs)) die("upload");
    bamm_packed_free(pk);

    bamm_em_params p;
    bamm_em_default_params(&p);
    p.K = (uint32_t)K_; p.W = (uint32_t)W_; p.bg_order = (uint32_t)bgModel_->getOrder(); p.q = q_;
    p.optimize_q = optimizeQ_ ? 1 : 0;
    p.epsilon = epsilon_; p.max_iterations = (uint32_t)maxEMIterations_;
    std::vector<float> vbg, A, v;                        // flat layouts of include/bamm_em.h
    for (size_t k = 0; k <= bgModel_->getOrder(); k++) vbg.insert(vbg.end(), bgModel_->getV()[k], bgModel_->getV()[k] + Y_[k + 1]);
    for (size_t k = 0; k <= K_; k++) A.insert(A.end(), A_[k], A_[k] + W_);
    for (size_t k = 0; k <= K_; k++)
        for (size_t y = 0; y < Y_[k + 1]; y++) v.insert(v.end(), motif_->getV()[k][y], motif_->getV()[k][y] + W_);
    if (bamm_em_create(d.ctx, d.seqs, &p, vbg.data(), A.data(), v.data(), nullptr, &d.em)) die("EM");
}

EM::~EM() {
    Device d;
    {
        std::lock_guard<std::mutex> lock(g_mu);
        d = g_dev[this];
        g_dev.erase(this);
    }
    bamm_em_destroy(d.em);
    bamm_seqs_destroy(d.seqs);
    bamm_ctx_destroy(d.ctx);
    if (r_) {
        for (size_t n = 0; n < seqs_.size(); n++) free(r_[n]);
        free(r_);
    }
    for (size_t k = 0; k < K_ + 1; k++) {
        for (size_t y = 0; y < Y_[k + 1]; y++) free(n_[k][y]);
        free(n_[k]);
    }
    free(n_);
}

namespace {

// device results back into the caller's objects: Motif::v_ (what Motif::updateV wrote in the reference),
// Motif::s_ as the last E pass saw it, EM::n_, q_, llikelihood_
void pull(EM* self, Device& d, Motif* motif, size_t K, size_t W, const std::vector<size_t>& Y, float*** n, float& q, float& llh,
          bool model_changed) {
    if (model_changed) {
        std::vector<float> v(bamm_v_size((uint32_t)K, (uint32_t)W)), cnt(v.size());
        if (bamm_em_get_v(d.em, v.data()) || bamm_em_get_counts(d.em, cnt.data())) die("read-back");
        size_t o = 0;
        for (size_t k = 0; k <= K; k++)
            for (size_t y = 0; y < Y[k + 1]; y++)
                for (size_t j = 0; j < W; j++, o++) { motif->getV()[k][y][j] = v[o]; n[k][y][j] = cnt[o]; }
    }
    if (bamm_em_get_q(d.em, &q) || bamm_em_get_llh(d.em, &llh)) die("read-back");
    (void)self;
}

}  // namespace

void EM::EStep() {                                       // EM.cpp:139-200
    Device& d = dev(this);
    if (bamm_em_estep(d.em)) die("EStep");
    std::vector<float> s(Y_[K_ + 1] * W_);
    if (bamm_em_get_s(d.em, s.data())) die("read-back");  // Motif::calculateLinearS (EM.cpp:143) left this in the Motif
    for (size_t y = 0; y < Y_[K_ + 1]; y++)
        for (size_t j = 0; j < W_; j++) s_[y][j] = s[y * W_ + j];
    d.r_fresh = false;
    pull(this, d, motif_, K_, W_, Y_, n_, q_, llikelihood_, false);
}

void EM::MStep() {                                       // EM.cpp:217-259
    Device& d = dev(this);
    if (bamm_em_mstep(d.em)) die("MStep");
    pull(this, d, motif_, K_, W_, Y_, n_, q_, llikelihood_, true);
}

void EM::optimize_q() {                                  // EM.cpp:505-519
    Device& d = dev(this);
    if (bamm_em_optimize_q(d.em)) die("optimize_q");
    if (bamm_em_get_q(d.em, &q_)) die("read-back");
}

float EM::getQ() { return q_; }

float** EM::getR() {                                     // EM.cpp:521: r_[n][L-W-i] for window start i
    Device& d = dev(this);
    if (!r_) {
        r_ = (float**)calloc(seqs_.size(), sizeof(float*));
        for (size_t n = 0; n < seqs_.size(); n++) r_[n] = (float*)calloc(seqs_[n]->getL(), sizeof(float));
    }
    if (!d.r_fresh) {
        std::vector<float> flat(d.total_positions ? d.total_positions : 1);
        if (bamm_em_get_r(d.em, 0, seqs_.size(), flat.data(), d.total_positions)) die("getR");
        size_t o = 0;
        for (size_t n = 0; n < seqs_.size(); n++) {
            memcpy(r_[n], flat.data() + o, seqs_[n]->getL() * sizeof(float));
            o += seqs_[n]->getL();
        }
        d.r_fresh = true;
    }
    return r_;
}

static int run(EM* self, Device& d, bool masked, float f, bool verbose, bool optimizeQ) {
    auto t0 = std::chrono::high_resolution_clock::now();
    uint32_t it = 0;
    if (masked ? bamm_em_mask(d.em, f, &it, nullptr, nullptr) : bamm_em_optimize(d.em, &it)) die(masked ? "mask" : "optimize");
    d.r_fresh = false;
    if (verbose) {                                       // the lines EM.cpp:112-115 / :487 print
        std::vector<float> llh(it), vd(it), q(it);
        uint32_t n = 0;
        bamm_em_get_trace(d.em, llh.data(), vd.data(), q.data(), it, &n);
        for (uint32_t i = 0; i < n && i < it; i++) {
            if (masked) { std::cout << i + 1 << "th iteration, delta_llikelihood=" << llh[i] - (i ? llh[i - 1] : 0.f) << std::endl; continue; }
            if (optimizeQ && i < 5) std::cout << "optimized q=" << q[i] << std::endl;
            std::cout << i + 1 << " iter, llh=" << llh[i] << ", diff_llh=" << llh[i] - (i ? llh[i - 1] : 0.f)
                      << ", v_diff=" << vd[i] << std::endl;
        }
    }
    (void)self;
    auto dt = std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - t0);
    std::cout << "\n--- Runtime for EM: " << dt.count() << " seconds ---\n";          // EM.cpp:134
    return 0;
}

int EM::optimize() {                                     // EM.cpp:62-137
    Device& d = dev(this);
    run(this, d, false, f_, verbose_, optimizeQ_);
    pull(this, d, motif_, K_, W_, Y_, n_, q_, llikelihood_, true);
    motif_->calculateP();                                // EM.cpp:131
    return 0;
}

int EM::mask() {                                         // EM.cpp:261-503
    Device& d = dev(this);
    run(this, d, true, f_, verbose_, optimizeQ_);
    pull(this, d, motif_, K_, W_, Y_, n_, q_, llikelihood_, true);
    motif_->calculateP();                                // EM.cpp:499
    return 0;
}

void EM::print() {                                       // EM.cpp:529-539
    for (size_t j = 0; j < W_; j++) {
        for (size_t y = 0; y < Y_[K_ + 1]; y++) std::cout << std::setprecision(3) << n_[K_][y][j] << '\t';
        std::cout << std::endl;
    }
}

void EM::printR() {                                      // EM.cpp:541-551
    getR();
    for (size_t n = 0; n < seqs_.size(); n++) {
        std::cout << "seq " << n << ":" << std::endl;
        for (size_t i = 0; i + W_ <= seqs_[n]->getL(); i++) std::cout << r_[n][seqs_[n]->getL() - W_ - i] << '\t';
        std::cout << std::endl;
    }
}

void EM::write(char* odir, std::string basename, bool ss) {   // EM.cpp:553-615: .counts and .positions
    getR();
    const std::string opath = std::string(odir) + '/' + basename;
    std::ofstream ofile_n((opath + ".counts").c_str());
    for (size_t j = 0; j < W_; j++) {
        for (size_t k = 0; k < K_ + 1; k++) {
            for (size_t y = 0; y < Y_[k + 1]; y++) ofile_n << static_cast<int>(n_[k][y][j]) << '\t';
            ofile_n << std::endl;
        }
        ofile_n << std::endl;
    }
    std::ofstream ofile_pos((opath + ".positions").c_str());
    ofile_pos << "seq\tlength\tstrand\tstart..end\tpattern" << std::endl;
    const float cutoff = 0.3f;
    for (size_t n = 0; n < seqs_.size(); n++) {
        size_t L = seqs_[n]->getL();
        L = ss ? L : (L - 1) / 2;
        for (size_t i = 0; i + W_ <= seqs_[n]->getL(); i++) {
            if (r_[n][seqs_[n]->getL() - W_ - i] >= cutoff) {
                ofile_pos << seqs_[n]->getHeader() << '\t' << L << '\t' << ((i < L) ? '+' : '-') << '\t' << i + 1 << ".." << i + W_ << '\t';
                for (size_t b = i; b < i + W_; b++) ofile_pos << Alphabet::getBase(seqs_[n]->getSequence()[b]);
                ofile_pos << std::endl;
            }
        }
    }
}
