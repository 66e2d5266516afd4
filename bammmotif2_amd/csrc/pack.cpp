// Host-only part of the C ABI: 2-bit packing of the reference's kmer_ arrays, sharding and the
// small Motif helpers.  No HIP calls in this file -- it is exercised by the CPU test-suite.
//
// Reference behaviour restated (not copied): Sequence::Sequence builds kmer_[i] from up to 11
// bases ending at i, newest base = least-significant base-4 digit, and replaces an unknown
// base by rand()%4 independently for every (position, digit) term
// (/root/reference/src/init/Sequence.cpp:35-41).

#include <algorithm>
#include <atomic>
#include <cmath>
#include <thread>
#include <cstdarg>
#include <cstdlib>
#include <cstring>
#include <new>

#include "common.h"
#include "glibc_rand.h"

namespace bamm {

static thread_local std::string g_last_error;

void set_error(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
}

}  // namespace bamm

using namespace bamm;

namespace bamm {
void bg_from_top_counts(const uint64_t* top_counts, uint32_t K, const float* alpha, float* vbg_out);
}

namespace {

std::atomic<uint32_t> g_host_threads{0};

uint32_t host_threads() {
    uint32_t n = g_host_threads.load();
    if (n) return n;
    n = std::thread::hardware_concurrency();
    return std::max(1u, std::min(n ? n : 1u, 8u));      // unset: a modest default, the CLI passes --threads
}

// fn(thread, begin, end) over contiguous, ordered ranges of [0, n)
template <class F>
void parallel_ranges(uint64_t n, uint32_t T, F&& fn) {
    if (T <= 1 || n < 2) { fn(0u, uint64_t(0), n); return; }
    std::vector<std::thread> th;
    th.reserve(T);
    for (uint32_t t = 0; t < T; t++) {
        const uint64_t a = n * t / T, b2 = n * (t + 1) / T;
        th.emplace_back([&fn, t, a, b2]() { fn(t, a, b2); });
    }
    for (auto& x : th) x.join();
}

}  // namespace

extern "C" {

const char* bamm_last_error(void) { return g_last_error.c_str(); }
const char* bamm_version(void) { return "bammmotif2_amd 0.1 (gfx950)"; }

size_t bamm_v_size(uint32_t K, uint32_t W) { return v_size(K, W); }
size_t bamm_v_offset(uint32_t k, uint32_t W) { return v_offset(k, W); }
size_t bamm_bg_size(uint32_t K) { return bg_size(K); }

void bamm_packed_free(bamm_packed* p) {
    if (!p) return;
    free(p->words);
    free(p->word_off);
    free(p->len);
    free(p->exc_off);
    free(p->exc_pos);
    free(p->exc_kmer);
    free(p->exc_clean);
    free(p);
}

namespace {

struct PackBuilder {
    bamm_packed* p = nullptr;
    std::vector<uint32_t> epos, ekmer, eclean;

    int begin(uint64_t n_seqs, const std::vector<uint64_t>& lens) {
        p = (bamm_packed*)calloc(1, sizeof(bamm_packed));
        if (!p) { set_error("out of memory"); return BAMM_ERR_ARG; }
        p->n_seqs = n_seqs;
        p->word_off = (uint64_t*)calloc(n_seqs + 1, sizeof(uint64_t));
        p->exc_off = (uint64_t*)calloc(n_seqs + 1, sizeof(uint64_t));
        p->len = (uint32_t*)calloc(n_seqs ? n_seqs : 1, sizeof(uint32_t));
        uint64_t n_words = 0, total = 0;
        uint32_t max_len = 0, min_len = n_seqs ? UINT32_MAX : 0;
        for (uint64_t n = 0; n < n_seqs; n++) {
            const uint64_t L = lens[n];
            if (L > 0xffffffffull) {
                set_error("sequence %llu longer than 2^32-1", (unsigned long long)n);
                bamm_packed_free(p);
                p = nullptr;
                return BAMM_ERR_ARG;
            }
            p->len[n] = (uint32_t)L;
            p->word_off[n] = n_words;
            n_words += (L + 15) / 16;
            total += L;
            if (L > max_len) max_len = (uint32_t)L;
            if (L < min_len) min_len = (uint32_t)L;
        }
        p->word_off[n_seqs] = n_words;
        p->n_words = n_words;
        p->total_len = total;
        p->max_len = max_len;
        p->min_len = min_len;
        p->words = (uint32_t*)calloc(n_words ? n_words : 1, sizeof(uint32_t));
        return BAMM_OK;
    }

    // km[i] = kmer_[i] as the reference holds it (any multiple of 4^11 may be added)
    void add(uint64_t n, const uint64_t* km) {
        p->exc_off[n] = epos.size();
        add_to(n, km, epos, ekmer, eclean);
    }
    // the same with the exceptions appended to the caller's vectors (one set per worker thread);
    // the words of different sequences never overlap
    void add_to(uint64_t n, const uint64_t* km, std::vector<uint32_t>& epos, std::vector<uint32_t>& ekmer,
                std::vector<uint32_t>& eclean) const {
        const uint32_t MASK22 = (1u << 22) - 1u;
        const uint32_t L = p->len[n];
        uint32_t* w = p->words + p->word_off[n];
        uint32_t clean = 0;
        for (uint32_t i = 0; i < L; i++) {
            const uint32_t k22 = (uint32_t)(km[i] & MASK22);
            const uint32_t base = k22 & 3u;                       // digit 0 = base at i
            w[i >> 4] |= base << (30u - 2u * (i & 15u));          // big-endian inside the word
            clean = ((clean << 2) | base) & MASK22;               // what the stream alone implies
            if (clean != k22) {
                epos.push_back(i);
                ekmer.push_back(k22);
                eclean.push_back(clean);
            }
        }
    }

    bamm_packed* finish() {
        p->exc_off[p->n_seqs] = epos.size();
        p->n_exc = epos.size();
        const size_t ne = epos.size() ? epos.size() : 1;
        p->exc_pos = (uint32_t*)calloc(ne, sizeof(uint32_t));
        p->exc_kmer = (uint32_t*)calloc(ne, sizeof(uint32_t));
        p->exc_clean = (uint32_t*)calloc(ne, sizeof(uint32_t));
        if (!epos.empty()) {
            memcpy(p->exc_pos, epos.data(), epos.size() * sizeof(uint32_t));
            memcpy(p->exc_kmer, ekmer.data(), epos.size() * sizeof(uint32_t));
            memcpy(p->exc_clean, eclean.data(), epos.size() * sizeof(uint32_t));
        }
        return p;
    }
};

}  // namespace

static int pack_impl(const uint64_t* const* ptrs, const uint64_t* flat, const uint64_t* off_or_len,
                     uint64_t n_seqs, bamm_packed** out) {
    if (!out || (n_seqs && !off_or_len) || (n_seqs && !ptrs && !flat)) {
        set_error("bamm_pack_kmers: null argument");
        return BAMM_ERR_ARG;
    }
    std::vector<uint64_t> lens(n_seqs);
    for (uint64_t n = 0; n < n_seqs; n++) lens[n] = ptrs ? off_or_len[n] : off_or_len[n + 1] - off_or_len[n];
    PackBuilder b;
    int rc = b.begin(n_seqs, lens);
    if (rc) return rc;
    for (uint64_t n = 0; n < n_seqs; n++) b.add(n, ptrs ? ptrs[n] : flat + off_or_len[n]);
    *out = b.finish();
    return BAMM_OK;
}

void bamm_set_host_threads(uint32_t n) { g_host_threads.store(n); }

}  // extern "C"

namespace bamm {
uint32_t host_threads_hint() { return host_threads(); }
// out[d] = rand() % 4 for the first D draws of the libc stream as srand(seed) leaves it.  glibc's generator is restated
// (glibc_rand.h), checked once against the running libc, and every host thread jumps to its share of the draws; where
// libc is another generator the draws come from srand(seed) + rand(), one after the other.  Postcondition either way:
// libc's stream stands at srand(seed), not advanced (include/bamm_em.h: bamm_pack_codes_seeded).
void rand_draws_mod4(uint32_t seed, uint64_t D, uint8_t* out) {
    GlibcRandStream gen;
    gen.start(seed);
    if (gen.fast) {
        const uint32_t T = host_threads();
        parallel_ranges(D, (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(T, D / 65536 + 1)), [&](uint32_t, uint64_t d0, uint64_t d1) {
            GlibcRandStream mine = gen;
            mine.jump(d0);
            for (uint64_t d = d0; d < d1; d++) out[d] = (uint8_t)(mine.next_fast() % 4);
        });
    } else {
        srand(seed);
        for (uint64_t d = 0; d < D; d++) out[d] = (uint8_t)(rand() % 4);
    }
    srand(seed);
}
}  // namespace bamm

extern "C" {

int bamm_pack_kmers(const uint64_t* kmer, const uint64_t* off, uint64_t n_seqs, bamm_packed** out) {
    return pack_impl(nullptr, kmer, off, n_seqs, out);
}

int bamm_pack_kmer_ptrs(const uint64_t* const* kmer_ptrs, const uint64_t* L, uint64_t n_seqs,
                        bamm_packed** out) {
    return pack_impl(kmer_ptrs, nullptr, L, n_seqs, out);
}

// Sequence::Sequence restated (init/Sequence.cpp:4-43, :91-99; Alphabet.cpp:46-55): from the
// alphabet codes of a FASTA record to kmer_ -- reverse complement appended behind an N
// separator unless single_strand, an unknown base (code 0) randomised with libc rand()%4 once
// per (position, digit) term in the reference's visiting order (positions ascending, digits
// from the oldest base to the newest).  The complement table maps code 0 to the *byte* 'N'
// (78), so an N of the forward strand is NOT randomised on the reverse strand but enters the
// k-mer arithmetic as the digit 77; kept as is.  Only the 10 positions behind such a byte need
// the term-by-term path, everything else is a rolling 22-bit window.
static int pack_codes_impl(const uint8_t* codes, const uint64_t* off, uint64_t n_seqs, int single_strand, bool seeded,
                           uint32_t seed, bamm_packed** out);

int bamm_rand_stream_draws(uint32_t seed, uint64_t skip, int use_jump, uint32_t count, int32_t* out, int* matches_libc) {
    if (!out && count) { set_error("bamm_rand_stream_draws: null output"); return BAMM_ERR_ARG; }
    if (matches_libc) *matches_libc = GlibcRandStream::libc_is_this_generator() ? 1 : 0;
    if (use_jump < 0) {
        srand(seed);
        for (uint64_t k = 0; k < skip; k++) (void)rand();
        for (uint32_t k = 0; k < count; k++) out[k] = rand();
        return BAMM_OK;
    }
    GlibcRandStream g;
    g.seed(seed);
    if (use_jump) g.jump(skip);
    else for (uint64_t k = 0; k < skip; k++) (void)g.next_fast();
    for (uint32_t k = 0; k < count; k++) out[k] = g.next_fast();
    return BAMM_OK;
}

int bamm_pack_codes(const uint8_t* codes, const uint64_t* off, uint64_t n_seqs, int single_strand,
                    bamm_packed** out) {
    return pack_codes_impl(codes, off, n_seqs, single_strand, false, 0u, out);
}

int bamm_pack_codes_seeded(const uint8_t* codes, const uint64_t* off, uint64_t n_seqs, int single_strand, uint32_t seed,
                           bamm_packed** out) {
    return pack_codes_impl(codes, off, n_seqs, single_strand, true, seed, out);
}

static int pack_codes_impl(const uint8_t* codes, const uint64_t* off, uint64_t n_seqs, int single_strand, bool seeded,
                           uint32_t seed, bamm_packed** out) {
    if (!out || (n_seqs && (!codes || !off))) {
        set_error("bamm_pack_codes: null argument");
        return BAMM_ERR_ARG;
    }
    std::vector<uint64_t> lens(n_seqs);
    uint64_t maxL = 0;
    for (uint64_t n = 0; n < n_seqs; n++) {
        const uint64_t L0 = off[n + 1] - off[n];
        lens[n] = single_strand ? L0 : 2 * L0 + 1;
        if (lens[n] > maxL) maxL = lens[n];
    }
    PackBuilder b;
    int rc = b.begin(n_seqs, lens);
    if (rc) return rc;
    // The only serial part is the libc rand() stream: an unknown base at position z is drawn once per
    // (position, digit) term that sees it, i.e. min(11, L - z) times, in position order.  The draws of
    // the whole set are taken up front, in the reference's order; the encoding itself then runs on
    // several host threads, each sequence reading its own slice of the draws.
    const uint32_t T = std::max<uint32_t>(1, std::min<uint64_t>(host_threads(), n_seqs ? n_seqs : 1));
    std::vector<uint64_t> doff(n_seqs + 1, 0);
    parallel_ranges(n_seqs, T, [&](uint32_t, uint64_t n0, uint64_t n1) {
        for (uint64_t n = n0; n < n1; n++) {
            const uint8_t* c = codes + off[n];
            const uint64_t L0 = off[n + 1] - off[n], L = lens[n];
            uint64_t d = 0;
            for (const uint8_t* z = (const uint8_t*)memchr(c, 0, L0); z; z = (const uint8_t*)memchr(z + 1, 0, (size_t)(c + L0 - z - 1)))
                d += std::min<uint64_t>(11, L - (uint64_t)(z - c));
            if (!single_strand) d += std::min<uint64_t>(11, L - L0);   // the separator behind the forward strand
            doff[n + 1] = d;
        }
    });
    for (uint64_t n = 0; n < n_seqs; n++) doff[n + 1] += doff[n];
    std::vector<uint8_t> draws(doff[n_seqs] ? doff[n_seqs] : 1);
    // ... unless the caller names the seed the stream starts from (bamm_pack_codes_seeded: the reference seeds once,
    // mainBaMM.cpp:22, and reads its positives first): glibc's generator is restated (glibc_rand.h), checked against
    // the running libc, and every host thread jumps to its share of the draws -- 11 M draws at 1 M double-stranded
    // sequences were a tenth of a second in one thread.  libc's own stream is left freshly seeded.
    if (seeded) rand_draws_mod4(seed, doff[n_seqs], draws.data());
    else for (uint64_t d = 0; d < doff[n_seqs]; d++) draws[d] = (uint8_t)(rand() % 4);

    struct Local { std::vector<uint32_t> epos, ekmer, eclean; uint64_t n0 = 0, n1 = 0; };
    std::vector<Local> loc(T);
    std::vector<uint32_t> ecount(n_seqs ? n_seqs : 1, 0);
    parallel_ranges(n_seqs, T, [&](uint32_t t, uint64_t n0, uint64_t n1) {
        Local& me = loc[t];
        me.n0 = n0; me.n1 = n1;
        std::vector<uint8_t> seq(maxL + 1);
        std::vector<uint64_t> km(maxL + 1);
        for (uint64_t n = n0; n < n1; n++) {
            const uint8_t* c = codes + off[n];
            const uint64_t L0 = off[n + 1] - off[n], L = lens[n];
            const uint8_t* dr = draws.data() + doff[n];
            if (single_strand) {
                memcpy(seq.data(), c, L0);
            } else {
                for (uint64_t i = 0; i < L0; i++) {
                    seq[i] = c[i];
                    const uint8_t x = c[i];
                    seq[2 * L0 - i] = (x >= 1 && x <= 4) ? (uint8_t)(5 - x) : (uint8_t)'N';
                }
                seq[L0] = 0;
            }
            uint64_t roll = 0;
            int64_t special_until = -1;                      // last position still seeing a special byte
            for (uint64_t i = 0; i < L; i++) {
                const uint8_t x = seq[i];
                if (x == 0 || x > 4) special_until = (int64_t)i + 10;
                if ((int64_t)i <= special_until) {
                    uint64_t acc = 0;
                    for (uint64_t k = (i < 10 ? i + 1 : 11); k > 0; k--) {
                        const uint8_t cc = seq[i - k + 1];
                        const uint64_t digit = (cc == 0) ? (uint64_t)*dr++ : (uint64_t)(cc - 1);
                        acc += digit << (2 * (k - 1));
                    }
                    km[i] = acc;
                    roll = acc;
                } else {
                    roll = ((roll << 2) | (uint64_t)(x - 1)) & ((1ull << 22) - 1);
                    km[i] = roll;
                }
            }
            const size_t before = me.epos.size();
            b.add_to(n, km.data(), me.epos, me.ekmer, me.eclean);
            ecount[n] = (uint32_t)(me.epos.size() - before);
        }
    });
    for (uint64_t n = 0; n < n_seqs; n++) b.p->exc_off[n + 1] = b.p->exc_off[n] + ecount[n];
    for (uint32_t t = 0; t < T; t++) {                       // ranges are contiguous and in thread order
        b.epos.insert(b.epos.end(), loc[t].epos.begin(), loc[t].epos.end());
        b.ekmer.insert(b.ekmer.end(), loc[t].ekmer.begin(), loc[t].ekmer.end());
        b.eclean.insert(b.eclean.end(), loc[t].eclean.begin(), loc[t].eclean.end());
    }
    *out = b.finish();
    return BAMM_OK;
}

int bamm_unpack_y(const bamm_packed* p, uint32_t K, uint32_t* y_out) {
    if (!p || !y_out || K > BAMM_MAX_ORDER) {
        set_error("bamm_unpack_y: bad argument");
        return BAMM_ERR_ARG;
    }
    const uint32_t maskY = (uint32_t)(ipow4(K + 1) - 1);
    std::vector<uint64_t> pos0(p->n_seqs + 1, 0);             // first output cell of every sequence
    for (uint64_t n = 0; n < p->n_seqs; n++) pos0[n + 1] = pos0[n] + p->len[n];
    const uint32_t T = std::max<uint32_t>(1, std::min<uint64_t>(host_threads(), p->n_seqs / 1024 + 1));
    parallel_ranges(p->n_seqs, T, [&](uint32_t, uint64_t n0, uint64_t n1) {      // 80 M cells at 200k x 401: 0.13 s on one thread
        for (uint64_t n = n0; n < n1; n++) {
            const uint32_t* w = p->words + p->word_off[n];
            uint32_t* out = y_out + pos0[n];
            uint32_t roll = 0;
            for (uint32_t i = 0; i < p->len[n]; i++) {
                const uint32_t base = (w[i >> 4] >> (30u - 2u * (i & 15u))) & 3u;
                roll = (roll << 2) | base;
                out[i] = roll & maskY;
            }
            for (uint64_t e = p->exc_off[n]; e < p->exc_off[n + 1]; e++)
                out[p->exc_pos[e]] = p->exc_kmer[e] & maskY;
        }
    });
    return BAMM_OK;
}

int bamm_shard_range(const uint32_t* len, uint64_t n_seqs, uint32_t W, uint32_t rank, uint32_t world,
                     uint64_t* begin, uint64_t* end) {
    if (!begin || !end || world == 0 || rank >= world || (n_seqs && !len)) {
        set_error("bamm_shard_range: bad argument");
        return BAMM_ERR_ARG;
    }
    // contiguous ranges balanced by the number of windows sum(L-W+1)
    long double total = 0;
    for (uint64_t n = 0; n < n_seqs; n++) total += (len[n] >= W) ? (len[n] - W + 1) : 0;
    auto cut = [&](uint32_t r) -> uint64_t {
        if (r == 0) return 0;
        if (r >= world) return n_seqs;
        const long double target = total * r / world;
        long double acc = 0;
        for (uint64_t n = 0; n < n_seqs; n++) {
            if (acc >= target) return n;
            acc += (len[n] >= W) ? (len[n] - W + 1) : 0;
        }
        return n_seqs;
    };
    *begin = cut(rank);
    *end = cut(rank + 1);
    return BAMM_OK;
}

// BackgroundModel::BackgroundModel + calculateV restated (init/BackgroundModel.cpp:26-42,
// :441-473): every position counts once per order with its (k+1)-mer kmer_[i] mod 4^(k+1)
// (positions i<k see implicit zero digits), then interpolated conditionals.
int bamm_bg_model(const bamm_packed* p, uint32_t K, const float* alpha, float* vbg_out) {
    if (!p || !alpha || !vbg_out || K > BAMM_MAX_ORDER) {
        set_error("bamm_bg_model: bad argument");
        return BAMM_ERR_ARG;
    }
    const uint32_t maskK = (uint32_t)(ipow4(K + 1) - 1);
    std::vector<uint64_t> top(ipow4(K + 1), 0);               // counts of the highest order
    // integer counts: ranges of sequences on host threads, per-thread tables summed afterwards
    const uint32_t T = std::max<uint32_t>(1, std::min<uint64_t>(host_threads(), p->n_seqs / 1024 + 1));
    std::vector<std::vector<uint64_t>> part(T, std::vector<uint64_t>(ipow4(K + 1), 0));
    parallel_ranges(p->n_seqs, T, [&](uint32_t t, uint64_t n0, uint64_t n1) {
        std::vector<uint64_t>& mine = part[t];
        for (uint64_t n = n0; n < n1; n++) {
            const uint32_t* w = p->words + p->word_off[n];
            uint64_t e = p->exc_off[n];
            const uint64_t e1 = p->exc_off[n + 1];
            uint32_t roll = 0;
            for (uint32_t i = 0; i < p->len[n]; i++) {
                const uint32_t base = (w[i >> 4] >> (30u - 2u * (i & 15u))) & 3u;
                roll = (roll << 2) | base;
                uint32_t y = roll & maskK;
                if (e < e1 && p->exc_pos[e] == i) { y = p->exc_kmer[e] & maskK; e++; }
                mine[y]++;
            }
        }
    });
    for (uint32_t t = 0; t < T; t++)
        for (size_t y = 0; y < top.size(); y++) top[y] += part[t][y];
    bg_from_top_counts(top.data(), K, alpha, vbg_out);
    return BAMM_OK;
}

}  // extern "C"

namespace bamm {
// BackgroundModel::calculateV (BackgroundModel.cpp:441-473) from the counts of the highest order (where they were
// counted does not matter: host threads here, csrc/prep.hip on the device): lower orders by marginalisation, then the
// interpolated conditionals
void bg_from_top_counts(const uint64_t* top_counts, uint32_t K, const float* alpha, float* vbg_out) {
    std::vector<std::vector<uint64_t>> cnt(K + 1);
    // lower orders: kmer mod 4^(k+1) = y_K mod 4^(k+1)
    cnt[K].assign(top_counts, top_counts + ipow4(K + 1));
    for (uint32_t k = K; k > 0; k--) {
        cnt[k - 1].assign(ipow4(k), 0);
        for (size_t y = 0; y < ipow4(k + 1); y++) cnt[k - 1][y % ipow4(k)] += cnt[k][y];
    }
    uint64_t base_counts = 0;
    for (size_t y = 0; y < 4; y++) base_counts += cnt[0][y];
    for (size_t y = 0; y < 4; y++)
        vbg_out[y] = ((float)cnt[0][y] + alpha[0] * 0.25f) / ((float)base_counts + alpha[0]);
    for (uint32_t k = 1; k <= K; k++) {
        float* vk = vbg_out + bg_offset(k);
        const float* vk1 = vbg_out + bg_offset(k - 1);
        for (size_t y = 0; y < ipow4(k + 1); y++)
            vk[y] = ((float)cnt[k][y] + alpha[k] * vk1[y % ipow4(k)]) / ((float)cnt[k - 1][y / 4] + alpha[k]);
    }
}
}  // namespace bamm

extern "C" {

void bamm_em_default_params(bamm_em_params* p) {
    if (!p) return;
    memset(p, 0, sizeof(*p));
    p->K = 2;                 // Global.cpp:35
    p->W = 0;
    p->bg_order = 2;          // Global.cpp:47
    p->q = 0.3f;              // Global.cpp:52
    p->optimize_q = 0;
    p->epsilon = 0.01f;       // EM.h:62
    p->max_iterations = 1000; // EM.h:63
    p->n_seqs_global = 0;
}

// Motif::calculateP (init/Motif.cpp:430-469): joint probabilities from the conditionals; for
// j < k the left context comes from the background model.
int bamm_calculate_p(const float* v, const float* vbg, uint32_t bg_order, uint32_t K, uint32_t W,
                     float* p) {
    if (!v || !vbg || !p || K > BAMM_MAX_ORDER || W == 0) {
        set_error("bamm_calculate_p: bad argument");
        return BAMM_ERR_ARG;
    }
    for (uint32_t j = 0; j < W; j++)
        for (uint32_t y = 0; y < 4; y++) p[y * W + j] = v[y * W + j];
    for (uint32_t k = 1; k <= K; k++) {
        float* pk = p + v_offset(k, W);
        const float* pk1 = p + v_offset(k - 1, W);
        const float* vk = v + v_offset(k, W);
        for (size_t y = 0; y < ipow4(k + 1); y++) {
            for (uint32_t j = 0; j < k && j < W; j++) {
                float acc = 1.0f;
                for (uint32_t i = 0; i <= j; i++)              // motif part of the context
                    acc *= v[v_offset(k - i, W) + (y / ipow4(i)) * W + (j - i)];
                for (uint32_t i = j + 1; i <= k; i++) {        // part reaching left of the motif
                    if ((k - i) <= bg_order || k <= bg_order)
                        acc *= vbg[bg_offset(k - i) + y / ipow4(i)];
                    else
                        acc *= vbg[bg_offset(bg_order) + (y / 4) % ipow4(bg_order + 1)];
                }
                pk[y * W + j] = acc;
            }
            for (uint32_t j = k; j < W; j++) pk[y * W + j] = vk[y * W + j] * pk1[(y / 4) * W + j - 1];
        }
    }
    return BAMM_OK;
}

}  // extern "C"
