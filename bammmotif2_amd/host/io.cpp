// FASTA reader and model-file I/O of the BaMMmotif drop-in path.  See bamm_host.h.
#include <omp.h>
#include <sys/stat.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <random>
#include <sstream>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include "bamm_host.h"

namespace bammhost {

static inline size_t ipow4(size_t e) { return size_t(1) << (2 * e); }
static inline size_t voff(size_t k, size_t W) { return W * ((ipow4(k + 1) - 4) / 3); }
static inline size_t bgoff(size_t k) { return (ipow4(k + 1) - 4) / 3; }

std::string base_name(const std::string& path) {
    // text between the last '/' and the last '.' (refinement/utils.h:66-85)
    size_t start = 0, end = path.size();
    const size_t slash = path.find_last_of('/');
    if (slash != std::string::npos && slash != 0) start = slash + 1;
    const size_t dot = path.find_last_of('.');
    if (dot != std::string::npos && dot > 0) end = dot;
    if (end < start) end = path.size();
    return path.substr(start, end - start);
}

static inline uint8_t base_code(char c) {
    switch (c) {   // Alphabet.cpp:36-40, STANDARD: upper and lower case, everything else is N
        case 'A': case 'a': return 1;
        case 'C': case 'c': return 2;
        case 'G': case 'g': return 3;
        case 'T': case 't': return 4;
        default: return 0;
    }
}

int read_fasta(const std::string& path, FastaSet& out, std::string& err) {
    // the whole file in one read, then a walk over its lines with memchr: same rules as the
    // getline-based loop of SequenceSet::readFASTA (SequenceSet.cpp:67-225), without a string per line
    // the file mapped, not copied (a million records are 200 MB: the copy alone was a quarter of this function); a file
    // that cannot be mapped (a pipe, /dev/stdin) is read the old way
    struct Bytes {
        const char* p = nullptr; size_t n = 0; void* map = nullptr; std::vector<char> own;
        const char* data() const { return p; }
        size_t size() const { return n; }
        ~Bytes() { if (map) munmap(map, n); }
    } buf;
    {
        const int fd = open(path.c_str(), O_RDONLY);
        if (fd < 0) { err = "Error: Cannot open FASTA file: " + path; return 1; }
        struct stat sb;
        void* m = MAP_FAILED;
        if (fstat(fd, &sb) == 0 && S_ISREG(sb.st_mode) && sb.st_size > 0)
            m = mmap(nullptr, (size_t)sb.st_size, PROT_READ, MAP_PRIVATE | MAP_POPULATE, fd, 0);
        if (m != MAP_FAILED) {
            buf.map = m; buf.p = (const char*)m; buf.n = (size_t)sb.st_size;
            (void)madvise(m, buf.n, MADV_SEQUENTIAL);
        } else {
            char chunk[1 << 16];
            ssize_t got;
            while ((got = read(fd, chunk, sizeof chunk)) > 0) buf.own.insert(buf.own.end(), chunk, chunk + got);
            if (got < 0) { close(fd); err = "Error: Cannot open FASTA file: " + path; return 1; }
            buf.p = buf.own.data(); buf.n = buf.own.size();
        }
        close(fd);
    }
    uint8_t lut[256];
    for (int c = 0; c < 256; c++) lut[c] = base_code((char)c);
    // The records are independent: the buffer is cut at header lines into one range per granted core and every range is
    // walked by the same rules as one pass over the file would (1 M records: 0.26 s of a 0.9 s command when parsed by one
    // thread).  Two walks: the first only measures (records, bases), the second writes codes, offsets and headers
    // straight into their final places -- no per-range sets to join, every page of the result touched once, by the
    // thread that fills it.
    struct alignas(128) Part {                               // a line of its own per thread
        size_t records = 0, bases = 0;                       // of the non-empty records
        size_t counts[5] = {0, 0, 0, 0, 0};                  // [0] = unknown bases
        size_t max_len = 0, min_len = SIZE_MAX, empty_entries = 0;
        std::string err;
    };
    // sink(header, header_len, first_line, n_lines_end): one call per non-empty record; `write` = the second walk
    auto walk = [&](const char* p, const char* const endp, Part& part, bool write, size_t rec_at, size_t code_at) {
        const char* hdr = nullptr; size_t hdr_n = 0;          // header line of the record being read (without the TAB / CR tail)
        bool have_header = false;
        size_t L = 0;                                         // its bases so far
        uint8_t* dst = write ? out.codes.data() + code_at : nullptr;
        size_t rec = rec_at;
        auto flush = [&]() {
            if (!have_header) return;
            have_header = false;
            if (L == 0) { if (!write) part.empty_entries++; return; }
            if (write) {
                dst += L;
                out.off[rec + 1] = (size_t)(dst - out.codes.data());
                if (hdr_n) out.headers[rec].assign(hdr, hdr_n); else out.headers[rec] = ">";
                rec++;
            } else {
                part.records++; part.bases += L;
                part.max_len = std::max(part.max_len, L);
                part.min_len = std::min(part.min_len, L);
            }
        };
        while (p < endp) {
            const char* nl = (const char*)memchr(p, '\n', (size_t)(endp - p));
            const char* le = nl ? nl : endp;                 // line = [p, le)
            const size_t n = (size_t)(le - p);
            if (n != 0) {                                    // blank lines are skipped
                if (p[0] == '>') {
                    flush();
                    have_header = true;
                    L = 0;
                    if (n == 1) {
                        hdr = p; hdr_n = 0;
                    } else {                                 // up to the first TAB, then up to the first CR
                        const char* tab = (const char*)memchr(p, '\t', n);
                        const size_t h1 = tab ? (size_t)(tab - p) : n;
                        const char* cr = (const char*)memchr(p, '\r', h1);
                        hdr = p; hdr_n = cr ? (size_t)(cr - p) : h1;
                    }
                } else if (have_header) {
                    if (!write) {
                        if (memchr(p, ' ', n)) { part.err = "Error: FASTA sequence contains space character: " + path; return; }
                    } else {
                        uint8_t* d = dst + L;
                        size_t c0[5] = {0, 0, 0, 0, 0}, c1[5] = {0, 0, 0, 0, 0};   // two sets: no store-to-load chain from base to base
                        size_t i = 0;
                        for (; i + 2 <= n; i += 2) {
                            const uint8_t a = lut[(unsigned char)p[i]], b = lut[(unsigned char)p[i + 1]];
                            d[i] = a; d[i + 1] = b;
                            c0[a]++; c1[b]++;
                        }
                        if (i < n) { const uint8_t a = lut[(unsigned char)p[i]]; d[i] = a; c0[a]++; }
                        for (int c = 0; c < 5; c++) part.counts[c] += c0[c] + c1[c];
                    }
                    L += n;
                } else {
                    part.err = "Error: Wrong FASTA format: " + path;
                    return;
                }
            }
            p = nl ? nl + 1 : endp;
        }
        flush();
    };
    const size_t T = std::max<size_t>(1, std::min<size_t>((size_t)host_parallelism(), buf.size() / (size_t(4) << 20) + 1));
    std::vector<const char*> cut(T + 1, buf.data() + buf.size());
    cut[0] = buf.data();
    for (size_t t = 1; t < T; t++) {                         // the first header line at or behind the t-th share of the bytes
        const char* q = buf.data() + buf.size() * t / T;
        const char* const endq = buf.data() + buf.size();
        while (q < endq) {
            const char* nl = (const char*)memchr(q, '\n', (size_t)(endq - q));
            if (!nl || nl + 1 >= endq) { q = endq; break; }
            if (nl[1] == '>') { q = nl + 1; break; }
            q = nl + 1;
        }
        cut[t] = std::max(q, cut[t - 1]);
    }
    std::vector<Part> parts(T);
    out = FastaSet();
#pragma omp parallel for schedule(static, 1) num_threads((int)T)
    for (long t = 0; t < (long)T; t++) walk(cut[(size_t)t], cut[(size_t)t + 1], parts[(size_t)t], false, 0, 0);
    std::vector<size_t> code_at(T + 1, 0), rec_at(T + 1, 0);
    for (size_t t = 0; t < T; t++) {                         // the first error in file order is the one a single pass stops at
        Part& part = parts[t];
        for (size_t e = 0; e < part.empty_entries; e++) fprintf(stderr, "Warning: Ignore FASTA entry without sequence: %s\n", path.c_str());
        if (!part.err.empty()) { err = part.err; return 1; }
        code_at[t + 1] = code_at[t] + part.bases;
        rec_at[t + 1] = rec_at[t] + part.records;
    }
    out.codes.resize(code_at[T]);                            // not initialised (ByteVec): the second walk writes every byte
    out.off.resize(rec_at[T] + 1);
    out.off[0] = 0;
    out.headers.resize(rec_at[T]);
#pragma omp parallel for schedule(static, 1) num_threads((int)T)
    for (long t = 0; t < (long)T; t++) walk(cut[(size_t)t], cut[(size_t)t + 1], parts[(size_t)t], true, rec_at[(size_t)t], code_at[(size_t)t]);
    size_t max_len = 0, min_len = SIZE_MAX;
    size_t counts[5] = {0, 0, 0, 0, 0};
    for (auto& part : parts) {
        max_len = std::max(max_len, part.max_len);
        min_len = std::min(min_len, part.min_len);
        for (int c = 0; c < 5; c++) counts[c] += part.counts[c];
    }
    out.max_len = max_len;
    out.min_len = out.size() ? min_len : 0;
    const size_t sum = counts[1] + counts[2] + counts[3] + counts[4];
    for (int i = 0; i < 4; i++) out.base_freq[i] = (float)counts[i + 1] / (float)sum;
    return 0;
}

// ------------------------------------------------------------------------- background ----
int bg_learn(const bamm_packed* p, uint32_t K, const std::vector<float>& alpha, BgModel& out) {
    out.K = K;
    out.alpha = alpha;
    out.v.assign(bamm_bg_size(K), 0.f);
    return bamm_bg_model(p, K, alpha.data(), out.v.data());
}

int bg_read(const std::string& path, BgModel& out, std::string& err) {
    struct stat sb;
    FILE* f = fopen(path.c_str(), "r");
    if (!f || stat(path.c_str(), &sb) != 0 || !S_ISREG(sb.st_mode)) {
        if (f) fclose(f);
        err = "Error: Input Background Model file does not exist.";
        return 1;
    }
    int K = 0;
    float a = 0;
    const std::string bad = "Error: Wrong BaMM format: " + path;
    if (fscanf(f, "# K = %d\n", &K) != 1 || K < 0 || K > BAMM_MAX_ORDER) { fclose(f); err = bad; return 1; }
    out.K = (uint32_t)K;
    out.alpha.assign(K + 1, 0.f);
    if (fscanf(f, "# A = %e", &a) != 1) { fclose(f); err = bad; return 1; }
    out.alpha[0] = a;
    for (int k = 1; k <= K; k++) {
        if (fscanf(f, "%e", &a) != 1) { fclose(f); err = bad; return 1; }
        out.alpha[k] = a;
    }
    out.v.assign(bamm_bg_size(K), 0.f);
    for (size_t i = 0; i < out.v.size(); i++) {
        float x;
        if (fscanf(f, "%e", &x) == EOF) { fclose(f); err = bad; return 1; }
        out.v[i] = x;
    }
    fclose(f);
    return 0;
}

int bg_write(const std::string& dir, const std::string& basename, const BgModel& bg, std::string& err) {
    const uint32_t K = bg.K;
    {   // conditional probabilities (BackgroundModel.cpp:359-377)
        std::ofstream file(dir + '/' + basename + ".hbcp");
        if (!file.is_open()) { err = "Error: Cannot write into output directory: " + dir; return 1; }
        file << "# K = " << K << std::endl;
        file << "# A =";
        for (uint32_t k = 0; k <= K; k++) file << " " << bg.alpha[k];
        file << std::endl;
        for (uint32_t k = 0; k <= K; k++) {
            for (size_t y = 0; y < ipow4(k + 1); y++)
                file << std::scientific << std::setprecision(6) << bg.v[bgoff(k) + y] << " ";
            file << std::endl;
        }
    }
    // joint probabilities p[k][y] = v[k][y] * p[k-1][y/4]  (BackgroundModel.cpp:393-405)
    std::vector<float> p(bg.v.size());
    for (size_t y = 0; y < 4; y++) p[y] = bg.v[y];
    for (uint32_t k = 1; k <= K; k++)
        for (size_t y = 0; y < ipow4(k + 1); y++) p[bgoff(k) + y] = bg.v[bgoff(k) + y] * p[bgoff(k - 1) + y / 4];
    std::ofstream file(dir + '/' + basename + ".hbp");
    if (!file.is_open()) { err = "Error: Cannot write into output directory: " + dir; return 1; }
    file << "# K = " << K << std::endl;
    file << "# A =";
    for (uint32_t k = 0; k <= K; k++) file << std::fixed << std::setprecision(2) << " " << bg.alpha[k];
    file << std::endl;
    for (uint32_t k = 0; k <= K; k++) {
        for (size_t y = 0; y < ipow4(k + 1); y++)
            file << std::scientific << std::setprecision(6) << p[bgoff(k) + y] << " ";
        file << std::endl;
    }
    return 0;
}

// ------------------------------------------------------------------------------ motif ----
void motif_alloc(Motif& m, uint32_t W, uint32_t K, const std::vector<float>& alpha, float q) {
    m.W = W;
    m.K = K;
    m.q = q;
    m.alpha = alpha;
    m.A.assign((size_t)(K + 1) * W, 0.f);
    for (uint32_t k = 0; k <= K; k++)
        for (uint32_t j = 0; j < W; j++) m.A[(size_t)k * W + j] = alpha[k];
    m.v.assign(bamm_v_size(K, W), 0.f);
    m.p.assign(bamm_v_size(K, W), 0.f);
}

void motif_calculate_p(Motif& m, const BgModel& bg) {
    bamm_calculate_p(m.v.data(), bg.v.data(), bg.K, m.K, m.W, m.p.data());
}

// interpolated higher orders from integer counts (Motif.cpp:314-326 / :399-428)
static void orders_from_counts(Motif& m, const std::vector<int>& n, bool order0_from_counts, size_t C, const BgModel& bg) {
    const uint32_t W = m.W, K = m.K;
    if (order0_from_counts)
        for (size_t y = 0; y < 4; y++)
            for (uint32_t j = 0; j < W; j++)
                m.v[y * W + j] = ((float)n[y * W + j] + m.A[j] * bg.v[y]) / ((float)C + m.A[j]);
    for (uint32_t k = 1; k <= K; k++)
        for (size_t y = 0; y < ipow4(k + 1); y++) {
            const size_t y2 = y % ipow4(k), yk = y / 4;
            for (uint32_t j = 0; j < k && j < W; j++) m.v[voff(k, W) + y * W + j] = m.v[voff(k - 1, W) + y2 * W + j];
            for (uint32_t j = k; j < W; j++) {
                const float Ak = m.A[(size_t)k * W + j];
                m.v[voff(k, W) + y * W + j] = ((float)n[voff(k, W) + y * W + j] + Ak * m.v[voff(k - 1, W) + y2 * W + j]) /
                                             ((float)n[voff(k - 1, W) + yk * W + j - 1] + Ak);
            }
        }
}

int motif_init_from_pwm(Motif& m, const std::vector<float>& pwm, const BgModel& bg, const uint32_t* yK,
                        const uint64_t* off, size_t n_seqs, float q, const SeedDevice* dev, std::string& err,
                        std::vector<uint32_t>* z_out) {
    if (z_out) z_out->assign(n_seqs, 0);
    const uint32_t W = m.W, K = m.K;
    m.q = q;
    std::vector<int> n(bamm_v_size(K, W), 0);
    for (uint32_t j = 0; j < W; j++) {                       // floor at 1e-8, renormalise (Motif.cpp:205-220)
        float norm = 0.0f;
        for (size_t y = 0; y < 4; y++) {
            const float x = pwm[y * W + j];
            m.v[y * W + j] = ((double)x <= 1.e-8) ? (float)1.e-8 : x;
            norm += m.v[y * W + j];
        }
        for (size_t y = 0; y < 4; y++) m.v[y * W + j] /= norm;
    }
    std::vector<float> score(4 * (size_t)W);
    for (size_t y = 0; y < 4; y++)
        for (uint32_t j = 0; j < W; j++) score[y * W + j] = m.v[y * W + j] / bg.v[y];
    std::mt19937 rngx;                                        // default-seeded on purpose (Motif.cpp:237)
    if (dev) {
        // device path: only the uniform variates are serial.  std::discrete_distribution::operator()
        // draws exactly one std::generate_canonical<double,53> per call (libstdc++ bits/random.tcc),
        // and the reference calls it once per sequence with L >= W, in sequence order.
        std::vector<double> u(n_seqs ? n_seqs : 1, 0.0);
        for (size_t s = 0; s < n_seqs; s++)
            if (off[s + 1] - off[s] >= W) u[s] = std::generate_canonical<double, 53>(rngx);
        if (bamm_seed_from_pwm(dev->ctx, dev->seqs, K, W, score.data(), q, u.data(), n.data(), z_out ? z_out->data() : nullptr)) {
            err = std::string("PWM seeding on the device failed: ") + bamm_last_error();
            return 1;
        }
        orders_from_counts(m, n, false, 0, bg);
        motif_calculate_p(m, bg);
        return 0;
    }
    // The posteriors of different sequences are independent, the draws are not (one RNG stream, in
    // sequence order): posteriors are computed for a block of sequences in parallel, then sampled
    // serially -- same results as the reference's loop run with one thread.
    const size_t kBlock = 2048;
    std::vector<std::vector<float>> post(kBlock);
    for (size_t s0 = 0; s0 < n_seqs; s0 += kBlock) {
        const size_t s1 = std::min(n_seqs, s0 + kBlock);
#pragma omp parallel for schedule(dynamic, 16)
        for (size_t s = s0; s < s1; s++) {
            std::vector<float>& r = post[s - s0];
            const size_t L = off[s + 1] - off[s];
            if (L < W) { r.clear(); continue; }                   // Motif.cpp:240-248
            const size_t LW1 = L - W + 1;
            const uint32_t* km = yK + off[s];
            r.assign(LW1 + 1, 0.f);
            float normFactor = 0.0f;
            const float pos0 = 1.0f - q, pos1 = q / (float)LW1;
            for (size_t i = 1; i <= LW1; i++) {
                r[i] = 1.0f;
                for (uint32_t j = 0; j < W; j++) r[i] *= score[(km[i - 1 + j] % 4) * W + j];
                r[i] *= pos1;
                normFactor += r[i];
            }
            r[0] = pos0;
            normFactor += r[0];
            for (size_t i = 0; i <= LW1; i++) r[i] /= normFactor;
        }
        for (size_t s = s0; s < s1; s++) {
            const std::vector<float>& r = post[s - s0];
            if (r.empty()) continue;
            const uint32_t* km = yK + off[s];
            std::discrete_distribution<size_t> dist(r.begin(), r.end());
            const size_t z = dist(rngx);
            if (z_out) (*z_out)[s] = (uint32_t)z;
            if (z > 0)
                for (uint32_t k = 0; k <= K; k++)
                    for (uint32_t j = 0; j < W; j++) n[voff(k, W) + (km[z - 1 + j] % ipow4(k + 1)) * W + j]++;
        }
    }
    orders_from_counts(m, n, false, 0, bg);
    motif_calculate_p(m, bg);
    (void)err;
    return 0;
}

int motif_init_from_bamm(Motif& m, const std::string& path, uint32_t l_flank, uint32_t r_flank, const BgModel& bg,
                         std::string& err) {
    std::ifstream file(path.c_str());
    if (!file.is_open()) { err = "Error: Input BaMM file cannot be opened!"; return 1; }
    const uint32_t W = m.W, K = m.K;
    auto flank = [&](uint32_t j) {
        for (uint32_t k = 0; k <= K; k++)
            for (size_t y = 0; y < ipow4(k + 1); y++) m.v[voff(k, W) + y * W + j] = 0.25f;
    };
    for (uint32_t j = 0; j < l_flank; j++) flank(j);
    std::string line;
    for (uint32_t j = l_flank; j + r_flank < W; j++) {        // K+1 lines per position, then one empty line
        for (uint32_t k = 0; k <= K; k++) {
            std::getline(file, line);
            std::stringstream number(line);
            for (size_t y = 0; y < ipow4(k + 1); y++) number >> m.v[voff(k, W) + y * W + j];
        }
        std::getline(file, line);
    }
    for (uint32_t j = W - r_flank; j < W; j++) flank(j);
    motif_calculate_p(m, bg);
    return 0;
}

int motif_init_from_sites(Motif& m, const std::string& path, uint32_t l_flank, uint32_t r_flank, const BgModel& bg,
                          std::string& err) {
    std::ifstream file(path.c_str());
    if (!file.is_open()) { err = "Error: Cannot open binding sites file: " + path; return 1; }
    const uint32_t W = m.W, K = m.K;
    static const char bases[] = "NACGT";
    std::vector<int> n(bamm_v_size(K, W), 0);
    size_t C = 0;
    std::string site;
    while (std::getline(file, site).good()) {
        C++;
        for (uint32_t i = 0; i < l_flank; i++) site.insert(site.begin(), bases[(uint8_t)rand() % 4 + 1]);
        for (uint32_t i = 0; i < r_flank; i++) site.insert(site.end(), bases[(uint8_t)rand() % 4 + 1]);
        if (site.size() != W) {
            char buf[160];
            snprintf(buf, sizeof buf, "Error: Length of binding site on line %d differs.\nBinding sites should have the same length.", (int)C);
            err = buf;
            return 1;
        }
        if (site.size() < K + 1) { err = "Error: Length of binding site sequence is shorter than model order."; return 1; }
        for (uint32_t k = 0; k <= K; k++)
            for (uint32_t j = k; j < W; j++) {
                size_t y = 0;
                for (uint32_t a = 0; a <= k; a++) y += ipow4(a) * (size_t)(base_code(site[j - a]) - 1);
                n[voff(k, W) + y * W + j]++;
            }
    }
    orders_from_counts(m, n, true, C, bg);
    motif_calculate_p(m, bg);
    return 0;
}

int motif_write(const std::string& dir, const std::string& basename, const Motif& m, std::string& err) {
    std::ofstream fv(dir + '/' + basename + ".ihbcp"), fp(dir + '/' + basename + ".ihbp");
    if (!fv.is_open() || !fp.is_open()) { err = "Error: Cannot write into output directory: " + dir; return 1; }
    for (uint32_t j = 0; j < m.W; j++) {                      // Motif.cpp:531-546
        for (uint32_t k = 0; k <= m.K; k++) {
            for (size_t y = 0; y < ipow4(k + 1); y++) {
                fv << std::scientific << std::setprecision(3) << m.v[voff(k, m.W) + y * m.W + j] << ' ';
                fp << std::scientific << std::setprecision(3) << m.p[voff(k, m.W) + y * m.W + j] << ' ';
            }
            fv << std::endl;
            fp << std::endl;
        }
        fv << std::endl;
        fp << std::endl;
    }
    return 0;
}

// -------------------------------------------------------------------------------- seeds ----
int load_seeds(const std::string& path, const std::string& tag, uint32_t l_flank, uint32_t r_flank, uint32_t K,
               const std::vector<float>& alpha, size_t max_pwm, float glob_q, const BgModel& bg, const uint32_t* yK,
               const uint64_t* off, size_t n_seqs, SeedSet& out, std::string& err, const SeedDevice* dev) {
    out = SeedSet();
    std::ifstream file(path.c_str());
    if (tag == "bindingsites") {
        if (!file.good()) { err = "Error: Cannot open binding sites file: " + path; return 1; }
        std::string first;
        std::getline(file, first);
        Motif m;
        motif_alloc(m, (uint32_t)first.size() + l_flank + r_flank, K, alpha, glob_q);
        if (motif_init_from_sites(m, path, l_flank, r_flank, bg, err)) return 1;
        out.max_w = m.W;
        out.motifs.push_back(std::move(m));
        return 0;
    }
    if (tag == "PWM") {
        if (!file.good()) { err = "Error: Cannot open PWM file: " + path; return 1; }
        std::string line, row;
        while (std::getline(file, line)) {
            if (line.find("letter-probability matrix") == std::string::npos) continue;   // MotifSet.cpp:71
            size_t asize = 0, length = 0;
            float q = glob_q;
            { std::stringstream s(line.substr(line.find("h=") + 2)); s >> asize; }
            { std::stringstream s(line.substr(line.find("w=") + 2)); s >> length; }
            if (line.find("occur=") != std::string::npos) {
                std::stringstream s(line.substr(line.find("occur=") + 7));               // MotifSet.cpp:86 (+7 as is)
                s >> q;
            }
            if (asize != 4) { err = "Error: only the STANDARD alphabet (alength= 4) is supported: " + path; return 1; }
            length += l_flank + r_flank;
            Motif m;
            motif_alloc(m, (uint32_t)length, K, alpha, q);
            std::vector<float> pwm(4 * length, 0.25f);
            for (size_t j = l_flank; j + r_flank < length; j++) {
                if (!std::getline(file, row)) {
                    err = "Error: Cannot find any PWM in the MEME-format file: " + path + "\nPlease check the content of your input MEME file.";
                    return 1;
                }
                std::stringstream number(row);
                for (size_t y = 0; y < 4; y++) number >> pwm[y * length + j];
            }
            if (motif_init_from_pwm(m, pwm, bg, yK, off, n_seqs, q, dev, err)) return 1;
            out.max_w = std::max(out.max_w, m.W);
            out.motifs.push_back(std::move(m));
            if (out.motifs.size() >= max_pwm) break;
        }
        if (out.motifs.empty()) {
            err = "Error: Cannot find any PWM in the MEME-format file: " + path + "\nPlease check the version of your input MEME file.";
            return 1;
        }
        return 0;
    }
    if (tag == "BaMM") {
        if (!file.good()) { err = "Error: Cannot open BaMM file: " + path; return 1; }
        size_t model_length = 0, model_order = 0, check_lines = 0;
        std::string line;
        while (std::getline(file, line)) {                    // MotifSet.cpp:182-201
            if (line.empty()) {
                model_length++;
                if (model_length > 1 && check_lines != model_order) { err = "This is not a BaMM-format file: " + path; return 1; }
                check_lines = 0;
            } else if (model_length == 0) {
                model_order++;
            } else {
                check_lines++;
            }
        }
        model_order -= 1;
        if (model_order > 8) { err = "The input BaMM model order is too high: " + path; return 1; }
        Motif m;
        motif_alloc(m, (uint32_t)(model_length + l_flank + r_flank), K, alpha, glob_q);
        if (motif_init_from_bamm(m, path, l_flank, r_flank, bg, err)) return 1;
        out.max_w = m.W;
        out.motifs.push_back(std::move(m));
        return 0;
    }
    err = "Error: unknown initial model tag " + tag;
    return 1;
}

}  // namespace bammhost
