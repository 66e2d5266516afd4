// ScoreSeqSet_hip.cpp -- ScoreSeqSet::calcLogOdds (src/seq_scoring/ScoreSeqSet.cpp:25-67) on an MI355X through
// libbamm_em.so.  The one member function a maintainer replaces; ScoreSeqSet.h and the rest of
// ScoreSeqSet.cpp (calcPvalues, write, writeLogOdds, the getters) stay the reference's.  In the read-only
// reference tree the original body cannot be deleted, so `make -C oracle ref_hip` compiles ScoreSeqSet.cpp with
// -DcalcLogOdds=calcLogOdds_cpu_unused: its definition gets another name and this one takes its place.
#include "seq_scoring/ScoreSeqSet.h"

#include "bamm_em.h"

void ScoreSeqSet::calcLogOdds() {
    auto die = [](const char* what) {
        std::cerr << "Error: " << what << ": " << bamm_last_error() << std::endl;
        exit(1);
    };
    const size_t K = motif_->getK(), W = motif_->getW();
    const size_t K_bg = (bg_->getOrder() < K) ? bg_->getOrder() : K;
    motif_->calculateLogS(bg_->getV(), K_bg);            // callers read Motif::getS() afterwards (ScoreSeqSet.cpp:35)

    bamm_ctx* ctx = nullptr;
    bamm_seqs* dseqs = nullptr;
    if (bamm_ctx_create(0, nullptr, &ctx)) die("no usable MI355X");
    std::vector<const uint64_t*> km(seqSet_.size());
    std::vector<uint64_t> L(seqSet_.size()), off(seqSet_.size() + 1, 0);
    for (size_t n = 0; n < seqSet_.size(); n++) {
        km[n] = reinterpret_cast<const uint64_t*>(seqSet_[n]->getKmer());
        L[n] = seqSet_[n]->getL();
        off[n + 1] = off[n] + L[n] - W + 1;
    }
    bamm_packed* pk = nullptr;
    if (bamm_pack_kmer_ptrs(km.data(), L.data(), seqSet_.size(), &pk)) die("packing the sequences");
    if (bamm_seqs_upload(ctx, pk, 0, seqSet_.size(), &dseqs)) die("upload");
    bamm_packed_free(pk);

    std::vector<float> v, vbg;
    for (size_t k = 0; k <= K; k++)
        for (size_t y = 0; y < Y_[k + 1]; y++) v.insert(v.end(), motif_->getV()[k][y], motif_->getV()[k][y] + W);
    for (size_t k = 0; k <= bg_->getOrder(); k++) vbg.insert(vbg.end(), bg_->getV()[k], bg_->getV()[k] + Y_[k + 1]);
    std::vector<float> mops(off.back() ? off.back() : 1), zoops(seqSet_.size() ? seqSet_.size() : 1);
    std::vector<uint64_t> z(seqSet_.size() ? seqSet_.size() : 1);
    if (bamm_logodds(ctx, dseqs, (uint32_t)K, (uint32_t)W, (uint32_t)bg_->getOrder(), v.data(), vbg.data(), mops.data(), off.back(),
                     zoops.data(), z.data())) die("calcLogOdds");
    mops_scores_.resize(seqSet_.size());
    for (size_t n = 0; n < seqSet_.size(); n++) {         // appended, as the reference's push_back loops do
        mops_scores_[n].insert(mops_scores_[n].end(), mops.begin() + off[n], mops.begin() + off[n + 1]);
        zoops_scores_.push_back(zoops[n]);
        z_.push_back((size_t)z[n]);
    }
    bamm_seqs_destroy(dseqs);
    bamm_ctx_destroy(ctx);
}
