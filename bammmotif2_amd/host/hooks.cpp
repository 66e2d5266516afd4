// extern "C" test hooks over the C++ host code (libbamm_host.so), so that the CPU test-suite can
// drive the FASTA reader, the seeders and the model-file writers through ctypes.
#include <cstring>

#include <omp.h>

#include <charconv>
#include "bamm_host.h"

using namespace bammhost;

static thread_local std::string g_err;

extern "C" {

const char* bh_last_error(void) { return g_err.c_str(); }

// two-call protocol: first with codes == NULL to get sizes
int bh_read_fasta(const char* path, uint64_t* n_seqs, uint64_t* n_codes, uint8_t* codes, uint64_t* off, float* base_freq) {
    FastaSet fs;
    if (read_fasta(path, fs, g_err)) return 1;
    *n_seqs = fs.size();
    *n_codes = fs.codes.size();
    if (codes) {
        memcpy(codes, fs.codes.data(), fs.codes.size());
        memcpy(off, fs.off.data(), fs.off.size() * sizeof(uint64_t));
        if (base_freq) memcpy(base_freq, fs.base_freq, sizeof fs.base_freq);
    }
    return 0;
}

int bh_write_bg(const char* dir, const char* base, uint32_t K, const float* alpha, const float* v) {
    BgModel bg;
    bg.K = K;
    bg.alpha.assign(alpha, alpha + K + 1);
    bg.v.assign(v, v + bamm_bg_size(K));
    return bg_write(dir, base, bg, g_err);
}

int bh_read_bg(const char* path, uint32_t* K, float* alpha, float* v, uint32_t cap_k) {
    BgModel bg;
    if (bg_read(path, bg, g_err)) return 1;
    if (bg.K > cap_k) { g_err = "order too high for the caller's buffers"; return 1; }
    *K = bg.K;
    memcpy(alpha, bg.alpha.data(), bg.alpha.size() * sizeof(float));
    memcpy(v, bg.v.data(), bg.v.size() * sizeof(float));
    return 0;
}

int bh_write_motif(const char* dir, const char* base, uint32_t W, uint32_t K, const float* v, uint32_t bg_order, const float* vbg) {
    Motif m;
    std::vector<float> alpha(K + 1, 1.f);
    motif_alloc(m, W, K, alpha, 0.3f);
    m.v.assign(v, v + bamm_v_size(K, W));
    BgModel bg;
    bg.K = bg_order;
    bg.v.assign(vbg, vbg + bamm_bg_size(bg_order));
    motif_calculate_p(m, bg);
    return motif_write(dir, base, m, g_err);
}

// seeds: returns the number of motifs; v_out holds motif `index` (flat), w_out its width, q_out its q
static int load_seed_impl(const char* path, const char* tag, uint32_t l_flank, uint32_t r_flank, uint32_t K, const float* alpha,
                 uint64_t max_pwm, float glob_q, uint32_t bg_order, const float* vbg, const bamm_packed* packed,
                 uint32_t index, uint32_t* n_motifs, uint32_t* w_out, float* q_out, float* v_out, uint64_t v_cap,
                 const SeedDevice* dev) {
    BgModel bg;
    bg.K = bg_order;
    bg.v.assign(vbg, vbg + bamm_bg_size(bg_order));
    std::vector<uint32_t> yK(packed->total_len ? packed->total_len : 1);
    if (bamm_unpack_y(packed, K, yK.data())) { g_err = bamm_last_error(); return 1; }
    std::vector<uint64_t> off(packed->n_seqs + 1, 0);
    for (uint64_t n = 0; n < packed->n_seqs; n++) off[n + 1] = off[n] + packed->len[n];
    SeedSet seeds;
    std::vector<float> al(alpha, alpha + K + 1);
    if (load_seeds(path, tag, l_flank, r_flank, K, al, max_pwm, glob_q, bg, yK.data(), off.data(), packed->n_seqs, seeds, g_err, dev)) return 1;
    *n_motifs = (uint32_t)seeds.motifs.size();
    if (index >= seeds.motifs.size()) { g_err = "motif index out of range"; return 1; }
    const Motif& m = seeds.motifs[index];
    *w_out = m.W;
    *q_out = m.q;
    if (m.v.size() > v_cap) { g_err = "v buffer too small"; return 1; }
    memcpy(v_out, m.v.data(), m.v.size() * sizeof(float));
    return 0;
}

// Motif::initFromPWM on a caller-provided PWM ([4][W]): the model and the sampled site of every sequence, through
// the host path (std::mt19937 + std::discrete_distribution) or, with ctx/seqs, the device path
int bh_pwm_sites(const float* pwm, uint32_t W, uint32_t K, const float* alpha, uint32_t bg_order, const float* vbg,
                 const bamm_packed* packed, float q, float* v_out, uint32_t* z_out, bamm_ctx* ctx, bamm_seqs* seqs) {
    BgModel bg;
    bg.K = bg_order;
    bg.v.assign(vbg, vbg + bamm_bg_size(bg_order));
    Motif m;
    motif_alloc(m, W, K, std::vector<float>(alpha, alpha + K + 1), q);
    std::vector<uint64_t> off(packed->n_seqs + 1, 0);
    for (uint64_t n = 0; n < packed->n_seqs; n++) off[n + 1] = off[n] + packed->len[n];
    std::vector<uint32_t> yK(packed->total_len ? packed->total_len : 1), z;
    if (bamm_unpack_y(packed, K, yK.data())) { g_err = bamm_last_error(); return 1; }
    SeedDevice dev;
    dev.ctx = ctx; dev.seqs = seqs;
    if (motif_init_from_pwm(m, std::vector<float>(pwm, pwm + 4 * (size_t)W), bg, yK.data(), off.data(), packed->n_seqs, q,
                            ctx ? &dev : nullptr, g_err, &z)) return 1;
    memcpy(v_out, m.v.data(), m.v.size() * sizeof(float));
    memcpy(z_out, z.data(), z.size() * sizeof(uint32_t));
    return 0;
}

int bh_load_seed(const char* path, const char* tag, uint32_t l_flank, uint32_t r_flank, uint32_t K, const float* alpha,
                 uint64_t max_pwm, float glob_q, uint32_t bg_order, const float* vbg, const bamm_packed* packed,
                 uint32_t index, uint32_t* n_motifs, uint32_t* w_out, float* q_out, float* v_out, uint64_t v_cap) {
    return load_seed_impl(path, tag, l_flank, r_flank, K, alpha, max_pwm, glob_q, bg_order, vbg, packed, index, n_motifs,
                          w_out, q_out, v_out, v_cap, nullptr);
}

// the same with initFromPWM's pass over the sequences on the device (ctx / seqs: the uploaded `packed`)
int bh_load_seed_dev(const char* path, const char* tag, uint32_t l_flank, uint32_t r_flank, uint32_t K, const float* alpha,
                     uint64_t max_pwm, float glob_q, uint32_t bg_order, const float* vbg, const bamm_packed* packed,
                     uint32_t index, uint32_t* n_motifs, uint32_t* w_out, float* q_out, float* v_out, uint64_t v_cap,
                     bamm_ctx* ctx, bamm_seqs* seqs) {
    SeedDevice dev;
    dev.ctx = ctx; dev.seqs = seqs;
    return load_seed_impl(path, tag, l_flank, r_flank, K, alpha, max_pwm, glob_q, bg_order, vbg, packed, index, n_motifs,
                          w_out, q_out, v_out, v_cap, &dev);
}

// OpenMP threads of the host-side loops (the CLI's --threads)
void bh_set_threads(int n) { omp_set_num_threads(n > 0 ? n : 1); set_host_parallelism(n > 0 ? n : 1); }
void bh_auto_threads() { set_host_parallelism(0); }

const char* bh_base_name(const char* path) {
    g_err = base_name(path);
    return g_err.c_str();
}

}  // extern "C"

// ---- evaluation-side hooks ---------------------------------------------------------------
extern "C" {

// negatives for a packed (positive) set; two-call protocol (codes == NULL -> sizes only)
int bh_sample_negatives_strided(const bamm_packed* packed, uint32_t s_order, uint64_t m_fold, int generic, uint64_t keep_stride,
                                uint64_t* n_out, uint64_t* n_codes, uint8_t* codes, uint64_t* off);
int bh_sample_negatives(const bamm_packed* packed, uint32_t s_order, uint64_t m_fold, int generic, uint64_t* n_out,
                        uint64_t* n_codes, uint8_t* codes, uint64_t* off) {
    return bh_sample_negatives_strided(packed, s_order, m_fold, generic, 0, n_out, n_codes, codes, off);
}
// keep_stride > 1: only every keep_stride-th negative (what --FDR scores, FDR.cpp:58-60) is generated and returned
int bh_sample_negatives_strided(const bamm_packed* packed, uint32_t s_order, uint64_t m_fold, int generic, uint64_t keep_stride,
                                uint64_t* n_out, uint64_t* n_codes, uint8_t* codes, uint64_t* off) {
    static thread_local ByteVec c;
    static thread_local std::vector<uint64_t> o;
    if (!codes) {
        std::vector<uint32_t> ys(packed->total_len ? packed->total_len : 1);
        if (bamm_unpack_y(packed, s_order, ys.data())) { g_err = bamm_last_error(); return 1; }
        std::vector<uint64_t> poff(packed->n_seqs + 1, 0);
        for (uint64_t n = 0; n < packed->n_seqs; n++) poff[n + 1] = poff[n] + packed->len[n];
        if (sample_negatives(ys.data(), poff.data(), packed->n_seqs, s_order, m_fold, generic != 0, c, o, g_err, (size_t)keep_stride)) return 1;
        *n_out = o.size() - 1;
        *n_codes = c.size();
        return 0;
    }
    memcpy(codes, c.data(), c.size());
    memcpy(off, o.data(), o.size() * sizeof(uint64_t));
    return 0;
}

int bh_fdr_stats(const float* pos_max, uint64_t n_pos_max, const float* neg_max, uint64_t n_neg_max, const float* pos_all,
                 uint64_t n_pos_all, const float* neg_all, uint64_t n_neg_all, uint64_t posN, uint64_t negN, float q, int mops,
                 int zoops, int save_pvalues, const char* dir, const char* base) {
    FdrResult r;
    fdr_statistics(std::vector<float>(pos_max, pos_max + n_pos_max), std::vector<float>(neg_max, neg_max + n_neg_max),
                   std::vector<float>(pos_all, pos_all + n_pos_all), std::vector<float>(neg_all, neg_all + n_neg_all), posN, negN,
                   q, mops != 0, zoops != 0, save_pvalues != 0, r);
    return fdr_write(dir, base, r, posN, negN, mops != 0, zoops != 0, true, save_pvalues != 0, g_err);
}

// tests: format_g against std::to_chars (which is printf's %g) on n floats; returns the number of differing strings and
// the first offender's bits
uint64_t bh_format_g_check(const float* x, uint64_t n, int precision, uint32_t* first_bad_bits) {
    uint64_t bad = 0;
    char a[48], b[48];
    for (uint64_t i = 0; i < n; i++) {
        const size_t la = format_g(a, x[i], precision);
        const size_t lb = (size_t)(std::to_chars(b, b + sizeof b, x[i], std::chars_format::general, precision).ptr - b);
        if (la != lb || memcmp(a, b, la)) { if (!bad && first_bad_bits) memcpy(first_bad_bits, &x[i], 4); bad++; }
    }
    return bad;
}

// --saveLogOdds writers on caller-provided scores
int bh_fdr_logodds(const float* pos_max, uint64_t n_pos_max, const float* neg_max, uint64_t n_neg_max, const float* pos_all,
                   uint64_t n_pos_all, const float* neg_all, uint64_t n_neg_all, uint64_t posN, uint64_t negN, int mops, int zoops,
                   int ascending, const char* dir, const char* base) {
    return fdr_logodds_write(dir, base, std::vector<float>(pos_max, pos_max + n_pos_max), std::vector<float>(neg_max, neg_max + n_neg_max),
                             std::vector<float>(pos_all, pos_all + n_pos_all), std::vector<float>(neg_all, neg_all + n_neg_all), posN, negN,
                             mops != 0, zoops != 0, ascending != 0, g_err);
}

int bh_logodds_zoops(const char* dir, const char* base, const char* header_prefix, int number_headers, const uint8_t* codes,
                     const uint64_t* off, uint64_t n_seqs, int revcomp, int ss, uint32_t W, const float* zoops, const uint64_t* z) {
    std::vector<std::string> headers;
    for (uint64_t n = 0; n < n_seqs; n++) headers.push_back(number_headers ? header_prefix + std::to_string(n) : std::string(header_prefix));
    return logodds_zoops_write(dir, base, headers, codes, off, n_seqs, revcomp != 0, ss != 0, W, zoops, z, g_err);
}

int bh_mops_pvalues(const float* pos_scores, uint64_t n_pos, const float* neg_all, uint64_t n_neg, uint64_t posN, float* p_out,
                    float* e_out) {
    std::vector<float> p, e;
    mops_pvalues(pos_scores, n_pos, std::vector<float>(neg_all, neg_all + n_neg), posN, p, e);
    memcpy(p_out, p.data(), p.size() * sizeof(float));
    memcpy(e_out, e.data(), e.size() * sizeof(float));
    return 0;
}

int bh_occurrence(const char* dir, const char* base, const uint8_t* codes, const uint64_t* off, uint64_t n_seqs, int ss,
                  uint32_t W, const float* p, const float* e, float cutoff) {
    std::vector<std::string> headers;
    for (uint64_t n = 0; n < n_seqs; n++) headers.push_back("seq" + std::to_string(n));
    return occurrence_write(dir, base, headers, codes, off, n_seqs, ss != 0, W, p, e, cutoff, g_err);
}

}  // extern "C"
