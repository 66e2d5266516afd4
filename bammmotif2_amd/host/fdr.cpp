// Evaluation side of the BaMMmotif drop-in: negative-set sampler, FDR / PR statistics and window
// p-values.  Restated from the reference lines cited in bamm_host.h; fp32 expression order kept.
#include <omp.h>
#include <sched.h>

#include <algorithm>
#include <atomic>
#include <charconv>
#include <cassert>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <iomanip>
#include <thread>

#include "bamm_host.h"
#include "../csrc/glibc_rand.h"

namespace bammhost {

static inline size_t ipow4(size_t e) { return size_t(1) << (2 * e); }
static inline size_t bgoff(size_t k) { return (ipow4(k + 1) - 4) / 3; }

namespace {

struct NegSampler {
    uint32_t s;
    std::vector<float> v, range_bar, v_seq;      // flat [k][y]
    std::vector<size_t> n, n_seq;
    std::vector<float> A;

    explicit NegSampler(uint32_t s_order) : s(s_order) {
        const size_t tot = bgoff(s + 1);
        v.assign(tot, 0.f); range_bar.assign(tot, 0.f); v_seq.assign(tot, 0.f);
        n.assign(tot, 0); n_seq.assign(tot, 0);
        A.assign(s + 1, 20.f);                    // SeqGenerator.cpp:29-32
    }

    void kmer_frequency(const uint32_t* y_s, const uint64_t* off, size_t n_seqs) {   // :63-110
        std::fill(n.begin(), n.end(), 0);
        // integer counts: a histogram per host thread, summed -- the same numbers however the sequences are cut
        const int T = std::max(1, std::min<int>(host_parallelism(), (int)(n_seqs / 4096 + 1)));
        std::vector<std::vector<size_t>> part((size_t)T, std::vector<size_t>(n.size(), 0));
#pragma omp parallel for schedule(static) num_threads(T)
        for (long i = 0; i < (long)n_seqs; i++) {
            std::vector<size_t>& mine = part[(size_t)omp_get_thread_num()];
            const size_t L = off[i + 1] - off[i];
            const uint32_t* km = y_s + off[i];
            for (uint32_t k = 0; k <= s; k++)
                for (size_t j = k; j < L; j++) mine[bgoff(k) + km[j] % ipow4(k + 1)]++;
        }
        for (auto& pt : part)
            for (size_t c = 0; c < n.size(); c++) n[c] += pt[c];
        size_t norm = 0;
        for (size_t y = 0; y < 4; y++) norm += n[y];
        float sum = 0.0f;
        for (size_t y = 0; y < 4; y++) {
            v[y] = ((float)n[y] + A[0] * 0.25f) / ((float)norm + A[0]);
            sum += v[y];
            range_bar[y] = sum;
        }
        for (uint32_t k = 1; k <= s; k++) {
            sum = 0.f;
            for (size_t y = 0; y < ipow4(k + 1); y++) {
                const size_t yk = y / 4, y2 = y % ipow4(k);
                v[bgoff(k) + y] = ((float)n[bgoff(k) + y] + A[k] * v[bgoff(k - 1) + y2]) / ((float)n[bgoff(k - 1) + yk] + A[k]);
                if (y % 4 == 0) sum = 0.f;
                sum += v[bgoff(k) + y];
                range_bar[bgoff(k) + y] = sum;
            }
        }
    }

    void rescale(const uint32_t* km, size_t L) {   // :112-186, written for s = 2
        std::fill(n_seq.begin(), n_seq.end(), 0);
        for (uint32_t k = 0; k <= s; k++)
            for (size_t j = k; j < L; j++) n_seq[bgoff(k) + km[j] % ipow4(k + 1)]++;
        float sum = 0.f;
        for (size_t y = 0; y < 4; y++) {
            v_seq[y] = v[y];
            sum += v_seq[y];
            range_bar[y] = sum;
        }
        {
            const uint32_t k = 1;
            const size_t o1 = bgoff(1), o0 = bgoff(0);
            for (size_t y = 0; y < 16; y++) {
                const size_t y2 = y % 4;
                v_seq[o1 + y] = v[o1 + y] * ((float)n_seq[o1 + y] + A[k - 1] * v[o0 + y2]) / v[o0 + y2] / ((float)L + A[k - 1]);
            }
            float norm[4] = {0.f, 0.f, 0.f, 0.f};
            for (size_t y = 0; y < 16; y++) {
                const size_t yk = y / 4;
                v_seq[o1 + y] = ((float)n_seq[o1 + y] + A[k] * v_seq[o1 + y]) / ((float)n_seq[o0 + yk] + A[k]);
                norm[yk] += v_seq[o1 + y];
            }
            for (size_t y = 0; y < 16; y++) v_seq[o1 + y] /= norm[y / 4];
            for (size_t y = 0; y < 16; y++) {
                if (y % 4 == 0) sum = 0.0f;
                sum += v_seq[o1 + y];
                range_bar[o1 + y] = sum;
            }
        }
        {
            const uint32_t k = 2;
            const size_t o2 = bgoff(2), o1 = bgoff(1);
            for (size_t y = 0; y < 64; y++) {
                const size_t y2 = y % 16, yk = y / 4;
                v_seq[o2 + y] = ((float)n_seq[o2 + y] + A[k] * v_seq[o1 + y2]) / ((float)n_seq[o1 + yk] + A[k]);
                if (y % 4 == 0) sum = 0.0f;
                sum += v_seq[o2 + y];
                range_bar[o2 + y] = sum;
            }
        }
    }

    // glibc's rand() restated (csrc/glibc_rand.h): used only after its first draws have been checked against
    // srand(42)/rand() of the running libc, otherwise the sampler stays on rand(); jump(n) lets every host thread
    // start in the middle of the one stream.  Every later rand() consumer reseeds (FDR.cpp:153).
    bamm::GlibcRandStream stream;


    // F sequences of the same length from the same tables, drawn one after the other from the
    // stream (sequence f consumes draws [f*L, (f+1)*L)) but advanced position by position together:
    // the per-base dependency chain (context -> bars -> base -> context) of one sequence overlaps
    // with those of the others.
    // keep (nullable): keep[f] == 0 -> sequence f still consumes its L draws of the stream but is not generated; the
    // kept ones are written one after the other (FDR.cpp:58-60 only ever scores every cvFold-th negative)
    void draw_many(size_t L, size_t F, uint8_t* seqs, std::vector<float>& rnd, std::vector<size_t>& ctx, const uint8_t* keep = nullptr) {
        if (s >= L || F == 1) {
            for (size_t f = 0; f < F; f++) {
                if (keep && !keep[f]) { for (size_t k = 0; k < L; k++) (void)stream.next(); continue; }
                draw(L, seqs);
                seqs += L;
            }
            return;
        }
        rnd.resize(F * L);
        for (size_t k = 0; k < F * L; k++) rnd[k] = (float)stream.next() / (float)RAND_MAX;
        ctx.assign(F, 0);
        std::vector<uint8_t*>& dst = dst_scratch;
        dst.assign(F, nullptr);
        {
            uint8_t* at = seqs;
            for (size_t f = 0; f < F; f++) if (!keep || keep[f]) { dst[f] = at; at += L; }
        }
        for (size_t f = 0; f < F; f++) {               // the first s bases: lower-order bars (as in draw())
            if (!dst[f]) continue;
            uint8_t* seq = dst[f];
            const float* r = rnd.data() + f * L;
            for (uint8_t y = 0; y < 4; y++)
                if (r[0] <= range_bar[y]) { seq[0] = y + 1; break; }
            for (size_t i = 1; i < s; i++) {
                size_t yk = 0;
                for (size_t k = i; k > 0; k--) yk += (size_t)(seq[i - k] - 1) * ipow4(k);
                for (size_t y = yk, a = 1; y < yk + 4; y++, a++) {
                    seq[i] = (uint8_t)a;
                    if (r[i] <= range_bar[bgoff(i) + y]) break;
                }
            }
            for (size_t k = s; k > 0; k--) ctx[f] = ctx[f] * 4 + (size_t)(seq[s - k] - 1);
        }
        const size_t ctx_mask = ipow4(s) - 1;
        const float* bar = range_bar.data() + bgoff(s);
        for (size_t i = s; i < L; i++)
            for (size_t f = 0; f < F; f++) {
                if (!dst[f]) continue;
                const float* b4 = bar + ctx[f] * 4;
                const float random = rnd[f * L + i];
                const uint8_t a = (uint8_t)(1 + (random > b4[0]) + (random > b4[1]) + (random > b4[2]));
                dst[f][i] = a;
                ctx[f] = (ctx[f] * 4 + (size_t)(a - 1)) & ctx_mask;
            }
    }
    std::vector<uint8_t*> dst_scratch;

    void draw(size_t L, uint8_t* seq) {             // :222-283 == :296-341 (same sampling loop)
        float random = (float)stream.next() / (float)RAND_MAX;
        for (uint8_t y = 0; y < 4; y++)
            if (random <= range_bar[y]) { seq[0] = y + 1; break; }
        for (size_t i = 1; i < s && i < L; i++) {
            size_t yk = 0;
            for (size_t k = i; k > 0; k--) yk += (size_t)(seq[i - k] - 1) * ipow4(k);
            random = (float)stream.next() / (float)RAND_MAX;
            for (size_t y = yk, a = 1; y < yk + 4; y++, a++) {
                seq[i] = (uint8_t)a;
                if (random <= range_bar[bgoff(i) + y]) break;
            }
        }
        if (s >= L) return;
        // context of the previous s bases, rolled forward (yk = sum_k (seq[i-k]-1) * 4^k)
        const size_t ctx_mod = ipow4(s);
        size_t ctx = 0;
        for (size_t k = s; k > 0; k--) ctx = ctx * 4 + (size_t)(seq[s - k] - 1);
        const float* bar = range_bar.data() + bgoff(s);
        for (size_t i = s; i < L; i++) {
            const float* b4 = bar + ctx * 4;
            random = (float)stream.next() / (float)RAND_MAX;
            // first base whose cumulative bar reaches `random`, the last one when none does; the
            // bars are running sums (non-decreasing), so counting the bars below is the same thing
            const uint8_t a = (uint8_t)(1 + (random > b4[0]) + (random > b4[1]) + (random > b4[2]));
            seq[i] = a;
            ctx = (ctx * 4 + (size_t)(a - 1)) & (ctx_mod - 1);
        }
    }
};

}  // namespace

// Host threads for work whose result does not depend on how it is cut (this sampler, packing, sorts): every core
// the process may use -- the affinity mask capped by the cgroup CPU quota -- whatever --threads says (the
// reference's flag sized its OpenMP EM loops, Global.cpp:331-333; a container may show 256 CPUs and grant 16).
static std::atomic<int> g_parallelism_override{0};
void set_host_parallelism(int n) { g_parallelism_override.store(n > 0 ? n : 0); }
int host_parallelism() {
    if (const int forced = g_parallelism_override.load()) return forced;
    static const int n = [] {
        int c = (int)std::thread::hardware_concurrency();
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof set, &set) == 0) c = CPU_COUNT(&set);
        std::ifstream f("/sys/fs/cgroup/cpu.max");
        std::string quota;
        double period = 0;
        if (f >> quota >> period && quota != "max" && period > 0) c = std::min(c, std::max(1, (int)(std::stod(quota) / period + 0.5)));
        return std::max(1, std::min(c, 64));
    }();
    return n;
}

int sample_negatives(const uint32_t* y_s, const uint64_t* off, size_t n_seqs, uint32_t s_order, size_t m_fold,
                     bool generic, ByteVec& codes_out, std::vector<uint64_t>& off_out, std::string& err, size_t keep_stride) {
    if (!generic && s_order != 2) {
        err = "Error: the sequence-specific negative sampler is written for -s 2 (SeqGenerator.cpp:112-186); use --genericNeg";
        return 1;
    }
    NegSampler g(s_order);
    g.stream.start();                                  // SeqGenerator.cpp:35: the stream as srand(42) leaves it
    // libc itself is touched only where its rand() is not the restated generator (then the draws below come from it, one
    // thread): this function runs on a thread beside the main one in the CLI, and libc's stream is process-global
    if (!g.stream.fast) srand(42);
    g.kmer_frequency(y_s, off, n_seqs);
    // Every negative consumes exactly L draws of the one rand() stream, in order, so where each positive
    // sequence's draws start is known up front.  With the restated generator (stream.fast) the stream
    // is cut into contiguous ranges, every range jumps to its first draw (GlibcRandStream::jump) and the ranges
    // are sampled on separate threads -- same draws, same negatives.
    // keep_stride > 1: only the negatives idx = 0, stride, 2 stride, ... with idx + stride <= total are generated and
    // returned (the others still consume their draws): --FDR scores nothing else (FDR.cpp:58-60).
    const size_t total_neg = n_seqs * m_fold;
    auto kept = [&](size_t idx) { return keep_stride <= 1 || (idx % keep_stride == 0 && idx + keep_stride <= total_neg); };
    std::vector<uint64_t> d0(n_seqs + 1, 0);                 // first draw of positive i
    std::vector<uint64_t> c0(n_seqs + 1, 0), k0(n_seqs + 1, 0);   // first output byte / first kept negative of positive i
    for (size_t i = 0; i < n_seqs; i++) {
        const uint64_t L = off[i + 1] - off[i];
        size_t nk = 0;
        for (size_t f = 0; f < m_fold; f++) nk += kept(i * m_fold + f);
        d0[i + 1] = d0[i] + L * m_fold;
        c0[i + 1] = c0[i] + L * nk;
        k0[i + 1] = k0[i] + nk;
    }
    codes_out.resize((size_t)c0[n_seqs]);                     // every byte is written by the range that owns it
    off_out.assign((size_t)k0[n_seqs] + 1, 0);
    const int threads = host_parallelism();
    const bool parallel = g.stream.fast && threads > 1 && n_seqs >= 256;
    const size_t P = parallel ? (size_t)threads : 1;
    std::vector<size_t> first(P + 1, n_seqs);
    first[0] = 0;
    for (size_t p = 1; p < P; p++)                            // ranges of about equal numbers of draws
        first[p] = (size_t)(std::lower_bound(d0.begin(), d0.end(), d0[n_seqs] * p / P) - d0.begin());
    for (size_t p = 1; p <= P; p++) first[p] = std::max(first[p], first[p - 1]);
    first[P] = n_seqs;
    std::vector<decltype(g.stream)> start(P, g.stream);
    for (size_t p = 1; p < P; p++) start[p].jump(d0[first[p]]);   // nobody steps through the stream
#pragma omp parallel for schedule(static, 1) num_threads((int)P)
    for (long p = 0; p < (long)P; p++) {
        NegSampler w = g;                                      // own tables (rescale writes them), own stream position
        w.stream = start[(size_t)p];
        std::vector<float> rnd2;
        std::vector<size_t> ctx2;
        std::vector<uint8_t> keep_f(m_fold, 1);
        for (size_t i = first[(size_t)p]; i < first[(size_t)p + 1]; i++) {
            const size_t L = off[i + 1] - off[i];
            // the reference recomputes the rescaled tables for every fold (SeqGenerator.cpp:296); they
            // only depend on the positive sequence, so once per sequence gives the same tables
            if (!generic) w.rescale(y_s + off[i], L);
            size_t nk = 0;
            for (size_t f = 0; f < m_fold; f++) {
                keep_f[f] = kept(i * m_fold + f) ? 1 : 0;
                if (keep_f[f]) { off_out[(size_t)k0[i] + nk + 1] = c0[i] + L * (nk + 1); nk++; }
            }
            w.draw_many(L, m_fold, codes_out.data() + c0[i], rnd2, ctx2, keep_stride > 1 ? keep_f.data() : nullptr);
        }
    }
    return 0;
}

// Scores in the reference's order (std::sort with greater / less, FDR.cpp:158-159,203-204,288-289) on every granted core:
// sorted runs per thread, merged pairwise.  A sorted sequence of floats is the same whatever produced it (equal
// scores are indistinguishable -- the rare +0 / -0 pair aside, which no statistic tells apart).
void sort_scores(std::vector<float>& v, bool descending) {
    const size_t n = v.size();
    const size_t T = std::max<size_t>(1, std::min<size_t>((size_t)host_parallelism(), n / 100000 + 1));
    auto less = [descending](float a, float b) { return descending ? a > b : a < b; };
    if (T == 1) { std::sort(v.begin(), v.end(), less); return; }
    std::vector<size_t> cut(T + 1);
    for (size_t t = 0; t <= T; t++) cut[t] = n * t / T;
#pragma omp parallel for schedule(static, 1) num_threads((int)T)
    for (long t = 0; t < (long)T; t++) std::sort(v.begin() + cut[t], v.begin() + cut[t + 1], less);
    for (size_t w = 1; w < T; w *= 2) {                   // runs of w ranges -> runs of 2w
        const long pairs = (long)((T + 2 * w - 1) / (2 * w));
#pragma omp parallel for schedule(static, 1) num_threads((int)T)
        for (long p = 0; p < pairs; p++) {
            const size_t a = (size_t)p * 2 * w, m = std::min(T, a + w), b = std::min(T, a + 2 * w);
            if (m < b) std::inplace_merge(v.begin() + cut[a], v.begin() + cut[m], v.begin() + cut[b], less);
        }
    }
}

void fdr_statistics(std::vector<float> posMax, std::vector<float> negMax, std::vector<float> posAll,
                    std::vector<float> negAll, size_t posN, size_t negN, float q, bool mops, bool zoops,
                    bool with_pvalues, FdrResult& r) {
    r = FdrResult();
    const float mFold = (float)negN / (float)posN;
    srand(42);                                         // FDR.cpp:153
    if (mops) {                                        // FDR.cpp:156-196
        sort_scores(posAll, true);
        sort_scores(negAll, true);
        size_t ip = 0, in = 0;
        float E_TP = 0.0f;
        size_t idx_max = posN + negN;
        const size_t len_all = posAll.size() + negAll.size();
        for (size_t i = 0; i < len_all; i++) {
            // the reference indexes past the end of both vectors here (FDR.cpp:174); an exhausted
            // list is treated as "-inf" instead of reading stale heap memory
            const bool take_pos = (ip < posAll.size()) && (in >= negAll.size() || posAll[ip] > negAll[in]);
            if (take_pos) ip++; else in++;
            r.mops_tp.push_back((float)ip - (float)in / mFold);
            r.mops_fp.push_back((float)in / mFold);
            if (E_TP == r.mops_tp[i]) idx_max = i;
            if (E_TP < r.mops_tp[i]) E_TP = r.mops_tp[i];
        }
        for (size_t i = 0; i < idx_max && i < len_all; i++) {
            r.mops_fdr.push_back(r.mops_fp[i] / (r.mops_tp[i] + r.mops_fp[i]));
            r.mops_rec.push_back(r.mops_tp[i] / E_TP);
        }
        r.occ_mult = E_TP / (float)posN;
    }
    if (zoops) {                                       // FDR.cpp:199-275
        sort_scores(posMax, true);
        sort_scores(negMax, true);
        size_t ip = 0, in = 0, min_idx_pos = 0;
        const size_t posN_est = static_cast<size_t>(q * (float)posN);
        // The strided CV split (FDR.cpp:49-60) drops the last posN % cvFold positives and keeps
        // cvFold * floor(negN / cvFold) negative scores, so the lists can be shorter than posN / negN.  The
        // reference walks posN + negN steps regardless and reads past the end of both (FDR.cpp:227-239);
        // here the walk covers the scores that exist -- the same rows whenever the counts divide evenly.
        const size_t nP = posMax.size(), nN = negMax.size();
        size_t n_top = (size_t)std::fmin(100, negN / 10);
        if (n_top >= nN) n_top = nN ? nN - 1 : 0;
        float lambda = 1e-16f;
        for (size_t l = 0; l < n_top; l++) lambda += negMax[l] - negMax[n_top];
        lambda /= n_top;
        float Sl = 0.f;
        auto P = [&](size_t i) { return i < nP ? posMax[i] : -INFINITY; };
        auto N = [&](size_t i) { return i < nN ? negMax[i] : -INFINITY; };
        // the walk itself is a serial chain (which list the next score comes from, one rand() per tie): it only
        // records, per step, the score and the two counts; everything computed FROM those -- the p-value with its two
        // binary searches, FDR, recall: most of the time at 2.2 M steps -- is then filled in on every granted core by
        // the same expressions
        const size_t steps = nP + nN;
        std::vector<float> step_score(steps);
        std::vector<uint32_t> step_ip(steps), step_in(steps);
        for (size_t i = 0; i < steps; i++) {
            if ((P(ip) > N(in) || ip == 0 || in >= nN) && ip < nP) {
                Sl = posMax[ip];
                ip++;
            } else if (P(ip) == N(in) && rand() % 2 == 0 && ip < nP) {       // same evaluation order: rand() drawn on every tie
                Sl = posMax[ip];
                ip++;
            } else {
                Sl = negMax[in];                                             // in < nN: both lists cannot be exhausted inside the walk
                in++;
            }
            step_score[i] = Sl; step_ip[i] = (uint32_t)ip; step_in[i] = (uint32_t)in;
            if (ip == posN_est) min_idx_pos = i;
        }
        r.zoops_tp.resize(steps); r.zoops_fp.resize(steps); r.pn_pvalue.resize(steps);
        r.zoops_fdr.resize(steps); r.zoops_rec.resize(steps);
#pragma omp parallel for schedule(static) num_threads(host_parallelism())
        for (long ii = 0; ii < (long)steps; ii++) {
            const size_t i = (size_t)ii, in = step_in[i];
            const float Sl = step_score[i];
            const float TP = (float)step_ip[i], FP = (float)in / mFold;
            r.zoops_tp[i] = TP;
            r.zoops_fp[i] = FP;
            float p_value;
            if (nN && Sl <= negMax[n_top]) {
                auto lo = std::lower_bound(negMax.begin(), negMax.end(), Sl, std::greater<float>());
                auto up = std::upper_bound(negMax.begin(), negMax.end(), Sl, std::greater<float>());
                const float Sl_upper = (lo == negMax.begin()) ? Sl : *(lo - 1);
                // for scores at or below the lowest negative the reference dereferences end() (FDR.cpp:244);
                // with its vectors that is untouched zero-filled heap, so 0 is what it computes with
                const float Sl_lower = (up == negMax.end()) ? 0.0f : *up;
                p_value = (in + (Sl_upper - Sl) / (Sl_upper - Sl_lower + 1e-5)) / (float)negN;
            } else if (nN) {
                p_value = n_top * expf((negMax[n_top] - Sl) / lambda) / negN;
            } else {
                p_value = 1.0f;                                              // no negative score at all
            }
            r.pn_pvalue[i] = p_value;
            r.zoops_fdr[i] = FP / (TP + FP);
            r.zoops_rec[i] = TP / (float)posN;
        }
        r.occ_frac = r.zoops_fp.empty() ? 1.0f : 1.0f - r.zoops_fp[min_idx_pos] / (float)posN;
    }
    if (with_pvalues) {                                // FDR.cpp:278-333
        auto pv = [](std::vector<float>& pos, std::vector<float>& neg, std::vector<float>& out) {
            sort_scores(neg, false);
            sort_scores(pos, false);
            for (size_t i = 0; i < pos.size(); i++) {
                const size_t low = std::lower_bound(neg.begin(), neg.end(), pos[i]) - neg.begin();
                const size_t up = std::upper_bound(neg.begin(), neg.end(), pos[i]) - neg.begin();
                float p = 1.0f - (float)(up + low) / (2.0f * (float)neg.size());
                if (p < 1.e-6) p = 1.e-6;
                if (p > 1.0f) p = 1.0f;
                out.push_back(p);
            }
        };
        if (mops) pv(posAll, negAll, r.mops_pvalue);
        if (zoops) pv(posMax, negMax, r.zoops_pvalue);
    }
}

// `ostream << float` at precision P, i.e. printf's %.Pg, without the general-purpose machinery: the P significant digits
// come from one double multiplication (a float times a power of ten up to 1e22 is one rounding away from exact) and
// are used only when that rounding cannot have changed them -- the scaled value further than 1e-6 from a half; anything
// else (zero, non-finite, exponents beyond the exact powers, the rare near-tie) goes to std::to_chars, which IS %g.
// 11 M numbers per .stats file: most of what the file costs.
size_t format_g(char* out, float xf, int P) {
    static const double p10[] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
    auto slow = [&]() { return (size_t)(std::to_chars(out, out + 40, xf, std::chars_format::general, P).ptr - out); };
    if (P < 1 || P > 9 || !(xf == xf) || xf == 0.0f || std::isinf(xf)) return slow();
    char* o = out;
    double x = (double)xf;
    if (x < 0) { *o++ = '-'; x = -x; }
    int e2;
    (void)std::frexp(x, &e2);                                // x in [2^(e2-1), 2^e2)
    int e = (int)std::floor((e2 - 1) * 0.30102999566398120); // floor(log10(x)) or one below
    if (e < -15 || e > 15) return slow();
    auto ge_pow = [&](int k) { return k >= 0 ? x >= p10[k] : x * p10[-k] >= 1.0; };   // x >= 10^k (x * 10^-k exact enough: checked again below)
    if (ge_pow(e + 1)) e++;
    const int k = P - 1 - e;                                 // scale to P digits in front of the point
    if (k < -22 || k > 22) return slow();
    double v = k >= 0 ? x * p10[k] : x / p10[-k];
    const double lo = p10[P - 1], hi = p10[P];
    if (v < lo || v >= hi) {                                 // e was one off at a power of ten (x * 10^-k rounding): settle it exactly
        if (v < lo) { e--; } else { e++; }
        const int k2 = P - 1 - e;
        if (k2 < -22 || k2 > 22) return slow();
        v = k2 >= 0 ? x * p10[k2] : x / p10[-k2];
        if (v < lo || v >= hi) return slow();
    }
    uint64_t m = (uint64_t)v;
    const double frac = v - (double)m;
    if (std::fabs(frac - 0.5) < 1e-6) return slow();         // too close to a tie for one rounding to be trusted
    if (frac > 0.5 && ++m == (uint64_t)hi) { m = (uint64_t)lo; e++; }
    char dig[16];
    for (int i = P - 1; i >= 0; i--) { dig[i] = (char)('0' + m % 10); m /= 10; }
    int nd = P;
    while (nd > 1 && dig[nd - 1] == '0') nd--;               // %g drops trailing zeros
    if (e < -4 || e >= P) {                                  // d.ddde+XX
        *o++ = dig[0];
        if (nd > 1) { *o++ = '.'; memcpy(o, dig + 1, (size_t)nd - 1); o += nd - 1; }
        *o++ = 'e';
        int ae = e;
        if (ae < 0) { *o++ = '-'; ae = -ae; } else { *o++ = '+'; }
        *o++ = (char)('0' + ae / 10); *o++ = (char)('0' + ae % 10);
    } else if (e >= 0) {
        const int ip = e + 1;                                // digits in front of the point (<= P)
        for (int i = 0; i < ip; i++) *o++ = i < nd ? dig[i] : '0';
        if (nd > ip) { *o++ = '.'; memcpy(o, dig + ip, (size_t)(nd - ip)); o += nd - ip; }
    } else {
        *o++ = '0'; *o++ = '.';
        for (int i = 0; i < -e - 1; i++) *o++ = '0';
        memcpy(o, dig, (size_t)nd); o += nd;
    }
    return (size_t)(o - out);
}

namespace {

// rows of floats exactly as `ostream << float` prints them (printf %g at the stream's precision),
// collected in memory and written once: the reference ends every row with std::endl, i.e. one
// write() per line, which at 2.2 M rows costs more than computing the statistics
struct alignas(128) RowWriter {                          // a line of its own: the string's length changes with every number, and the threads' writers sit side by side
    std::string buf;
    int precision;
    explicit RowWriter(int prec = 6) : precision(prec) { buf.reserve(1 << 20); }
    void num(float x) {
        char tmp[48];
        buf.append(tmp, format_g(tmp, x, precision));
    }
    void tab() { buf.push_back('\t'); }
    void nl() { buf.push_back('\n'); }
    void flush_to(std::ofstream& f) { f.write(buf.data(), (std::streamsize)buf.size()); buf.clear(); }
    void maybe_flush(std::ofstream& f) { if (buf.size() > (1 << 20) - 256) flush_to(f); }
};

// rows [0, n) formatted by row(w, i) on every granted core -- rounds of one chunk per thread, the chunks written in
// order: the same bytes as one writer walking all rows (formatting 11 M numbers is most of what the files cost), and
// the threads' buffers are reused from round to round (fresh pages are slow to come by in a container)
template <class Row>
void write_rows(std::ofstream& f, size_t n, int precision, Row&& row) {
    constexpr size_t kChunk = 16384;
    const int T = (int)std::max<size_t>(1, std::min<size_t>((size_t)host_parallelism(), n / kChunk + 1));
    std::vector<RowWriter> part((size_t)T, RowWriter(precision));
    for (size_t base = 0; base < n; base += kChunk * (size_t)T) {
#pragma omp parallel for schedule(static, 1) num_threads(T)
        for (int t = 0; t < T; t++) {
            RowWriter& w = part[(size_t)t];
            const size_t a = std::min(n, base + kChunk * (size_t)t), b = std::min(n, a + kChunk);
            for (size_t i = a; i < b; i++) row(w, i);
        }
        for (auto& w : part) w.flush_to(f);
    }
}

}  // namespace

int fdr_write(const std::string& dir, const std::string& basename, const FdrResult& r, size_t posN, size_t negN,
              bool mops, bool zoops, bool save_prs, bool save_pvalues, std::string& err) {
    const std::string opath = dir + '/' + basename;
    if (save_prs) {
        if (zoops) {                                   // FDR.cpp:385-407
            std::ofstream f(opath + ".zoops.stats");
            if (!f.is_open()) { err = "Error: Cannot write into output directory: " + dir; return 1; }
            f << "TP" << '\t' << "FP" << '\t' << "FDR" << '\t' << "Recall" << '\t' << "p-value" << '\t'
              << (float)negN / (float)posN << '\t' << r.occ_frac << std::endl;
            write_rows(f, r.zoops_fdr.size(), 6, [&](RowWriter& w, size_t i) {
                w.num(r.zoops_tp[i]); w.tab(); w.num(r.zoops_fp[i]); w.tab(); w.num(r.zoops_fdr[i]); w.tab();
                w.num(r.zoops_rec[i]); w.tab(); w.num(r.pn_pvalue[i]); w.tab(); w.nl();
            });
        }
        if (mops) {                                    // FDR.cpp:409-425
            std::ofstream f(opath + ".mops.stats");
            f << "TP" << '\t' << "FP" << '\t' << "FDR" << '\t' << "Recall" << '\t' << r.occ_mult << std::endl;
            write_rows(f, r.mops_fdr.size(), 6, [&](RowWriter& w, size_t i) {
                w.num(r.mops_tp[i]); w.tab(); w.num(r.mops_fp[i]); w.tab(); w.num(r.mops_fdr[i]); w.tab();
                w.num(r.mops_rec[i]); w.tab(); w.nl();
            });
        }
    }
    if (save_pvalues) {                                // FDR.cpp:428-449, setprecision(3)
        if (zoops) {
            std::ofstream f(opath + ".zoops.pvalues");
            write_rows(f, r.zoops_pvalue.size(), 3, [&](RowWriter& w, size_t i) { w.num(r.zoops_pvalue[i]); w.nl(); });
        }
        if (mops) {
            std::ofstream f(opath + ".mops.pvalues");
            write_rows(f, r.mops_pvalue.size(), 3, [&](RowWriter& w, size_t i) { w.num(r.mops_pvalue[i]); w.nl(); });
        }
    }
    return 0;
}

// FDR::write, saveLogOdds_ branch (FDR.cpp:416-450): score i of the positives beside score
// i * negN / posN of the negatives.  The reference writes its vectors in whatever order the statistics left
// them: descending after calculatePR (FDR.cpp:158-159,203-204), ascending when calculatePvalues ran as well
// (FDR.cpp:288-289,310-311).  A negative index beyond the list (negN not a multiple of cvFold) ends the file:
// the reference reads past the end there.
int fdr_logodds_write(const std::string& dir, const std::string& basename, std::vector<float> posMax, std::vector<float> negMax,
                      std::vector<float> posAll, std::vector<float> negAll, size_t posN, size_t negN, bool mops, bool zoops,
                      bool ascending, std::string& err) {
    auto order = [&](std::vector<float>& v) {
        sort_scores(v, !ascending);
    };
    auto emit = [&](const std::string& path, std::vector<float>& pos, std::vector<float>& neg) {
        order(pos); order(neg);
        std::ofstream f(path);
        if (!f.is_open()) { err = "Error: Cannot write into output directory: " + dir; return 1; }
        f << "positive" << '\t' << "negative" << std::endl;
        RowWriter w(6);
        for (size_t i = 0; i < pos.size(); i++) {
            const size_t j = i * negN / posN;
            if (j >= neg.size()) break;
            w.num(pos[i]); w.tab(); w.num(neg[j]); w.nl();
            w.maybe_flush(f);
        }
        w.flush_to(f);
        return 0;
    };
    if (zoops && emit(dir + '/' + basename + ".zoops.logOdds", posMax, negMax)) return 1;
    if (mops && emit(dir + '/' + basename + ".mops.logOdds", posAll, negAll)) return 1;
    return 0;
}

// ScoreSeqSet::writeLogOdds (seq_scoring/ScoreSeqSet.cpp:293-331): the best window of every sequence.
// `revcomp`: the stored sequences carry the reverse strand behind an N (positives unless --ss); the sampled
// negatives never do, although the reference halves their length column too when --ss is absent.
int logodds_zoops_write(const std::string& dir, const std::string& basename, const std::vector<std::string>& headers,
                        const uint8_t* codes, const uint64_t* off, size_t n_seqs, bool revcomp, bool ss, uint32_t W,
                        const float* zoops, const uint64_t* z, std::string& err) {
    std::ofstream f(dir + '/' + basename + ".logOddsZoops");
    if (!f.is_open()) { err = "Error: Cannot write into output directory: " + dir; return 1; }
    f << "seq\tlength\tstrand\tstart..end\tpattern\tzoops_score" << std::endl;
    static const char B[] = "NACGT";
    for (size_t n = 0; n < n_seqs; n++) {
        const size_t L0 = off[n + 1] - off[n], L = revcomp ? 2 * L0 + 1 : L0;
        size_t seqlen = L;
        if (!ss) seqlen = (seqlen - 1) / 2;
        const uint8_t* c = codes + off[n];
        f << headers[n] << '\t' << seqlen << '\t' << ((z[n] < seqlen) ? '+' : '-') << '\t' << z[n] + 1 << ".." << z[n] + W << '\t';
        for (size_t m = z[n]; m < z[n] + W; m++) {
            char b = 'N';                                      // Sequence::getSequence(): forward, N, reverse complement
            if (m < L0) b = B[c[m] <= 4 ? c[m] : 0];
            else if (revcomp && m > L0 && m < L) { const uint8_t x = c[2 * L0 - m]; b = (x >= 1 && x <= 4) ? B[5 - x] : 'N'; }
            f << b;
        }
        f << '\t' << std::setprecision(3) << zoops[n] << std::endl;
    }
    return 0;
}

void mops_pvalues(const float* pos_scores, size_t n_pos_scores, std::vector<float> neg, size_t posN,
                  std::vector<float>& p_out, std::vector<float>& e_out) {
    const size_t negN = neg.size();                    // ScoreSeqSet.cpp:75-93
    const float eps = 1.0e-5;
    std::sort(neg.begin(), neg.end(), std::less<float>());
    const size_t nTop = std::min(100, (int)negN / 10);
    const float S_ntop = neg[nTop];
    float lambda = 0.f;
    for (size_t n = 0; n < nTop; n++) lambda += (neg[n] - S_ntop);
    lambda = lambda / (float)nTop;
    p_out.resize(n_pos_scores);
    e_out.resize(n_pos_scores);
    for (size_t i = 0; i < n_pos_scores; i++) {        // ScoreSeqSet.cpp:97-125
        const float Sl = pos_scores[i];
        const size_t FPl = neg.end() - std::upper_bound(neg.begin(), neg.end(), Sl);
        float p;
        if (FPl == negN) {
            p = 1.f;
        } else if (FPl < 10 && fabs(lambda) > eps) {
            p = float(nTop) / (float)negN * expf(-(Sl - S_ntop) / lambda);
        } else {
            const float SlHigher = neg[negN - FPl - 1], SlLower = neg[negN - FPl];
            p = ((float)FPl + (SlHigher - Sl + eps) / (SlHigher - SlLower + eps)) / (float)negN;
        }
        p_out[i] = p;
        e_out[i] = p * (float)posN;
    }
}

int occurrence_write(const std::string& dir, const std::string& basename, const std::vector<std::string>& headers,
                     const uint8_t* codes, const uint64_t* off, size_t n_seqs, bool ss, uint32_t W, const float* p,
                     const float* e, float cutoff, std::string& err) {
    std::ofstream f(dir + '/' + basename + ".occurrence");
    if (!f.is_open()) { err = "Error: Cannot write into output directory: " + dir; return 1; }
    f << "seq\tlength\tstrand\tstart..end\tpattern\tp-value\te-value" << std::endl;
    static const char B[] = "NACGT";
    size_t o = 0;
    for (size_t n = 0; n < n_seqs; n++) {
        const size_t L0 = off[n + 1] - off[n], L = ss ? L0 : 2 * L0 + 1, seqlen = ss ? L : (L - 1) / 2;
        const uint8_t* c = codes + off[n];
        auto base_at = [&](size_t b) -> char {             // Sequence::getSequence(): forward, N, reverse complement
            if (b < L0) return B[c[b] <= 4 ? c[b] : 0];
            if (ss || b == L0) return 'N';
            const uint8_t x = c[2 * L0 - b];
            return (x >= 1 && x <= 4) ? B[5 - x] : 'N';
        };
        const size_t LW1 = L - W + 1;
        for (size_t i = 0; i < LW1; i++) {
            if (p[o + i] < cutoff) {
                f << headers[n] << '\t' << seqlen << '\t' << ((i < seqlen) ? '+' : '-') << '\t' << i + 1 << ".." << i + W << '\t';
                for (size_t m = i; m < i + W; m++) f << base_at(m);
                f << '\t' << std::setprecision(3) << p[o + i] << '\t' << e[o + i] << std::endl;
            }
        }
        o += LW1;
    }
    return 0;
}

}  // namespace bammhost
