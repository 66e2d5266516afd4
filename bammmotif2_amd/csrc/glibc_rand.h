// glibc's default rand() restated: TYPE_3 additive feedback, o[k] = o[k-3] + o[k-31] over Z / 2^32, seeded by the
// Park-Miller LCG with 310 outputs discarded, rand() = o[k] >> 1.  The reference draws everything random from this
// ONE stream -- the N randomisation of Sequence::Sequence (init/Sequence.cpp:38), the negative sampler
// (seq_generator/SeqGenerator.cpp:35,222-341) -- so reproducing its bytes means reproducing the stream, and doing
// that on more than one thread (or on the device) means starting in the middle of it: the generator is linear,
// jump(n) multiplies the state by t^n mod (t^31 - t^28 - 1) in ~2 log2(n) polynomial products.
// The restatement is relied on only if the running libc's srand()/rand() IS this generator: checked once per process
// (libc_is_this_generator(): sequential draws and a jump against libc's own stream), never again -- so that threads
// beside the main one do not touch libc's process-global state on the fast path.
#pragma once
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <mutex>

namespace bamm {

struct GlibcRandStream {
    uint32_t r[34];
    int i = 0;
    bool fast = false;
    void seed(uint32_t sd) {
        int32_t x[344 + 34];
        x[0] = (int32_t)sd;
        for (int k = 1; k < 31; k++) {
            int64_t w = (16807LL * x[k - 1]) % 2147483647LL;
            if (w < 0) w += 2147483647LL;
            x[k] = (int32_t)w;
        }
        for (int k = 31; k < 34; k++) x[k] = x[k - 31];
        for (int k = 34; k < 344; k++) x[k] = (int32_t)((uint32_t)x[k - 31] + (uint32_t)x[k - 3]);
        for (int k = 0; k < 34; k++) r[k] = (uint32_t)x[344 - 34 + k];   // the last 34 words are the state
        i = 0;
    }
    inline int next_fast() {                   // o[k] = o[k-31] + o[k-3] over a ring of 34
        int a = i + 3, b = i + 31;
        a -= a >= 34 ? 34 : 0;
        b -= b >= 34 ? 34 : 0;
        const uint32_t v = r[a] + r[b];
        r[i] = v;
        i = i + 1 == 34 ? 0 : i + 1;
        return (int)(v >> 1);
    }
    inline int next() { return fast ? next_fast() : rand(); }
    static void poly_mul(const uint32_t* a, const uint32_t* b, uint32_t* out) {   // out = a * b mod (t^31 - t^28 - 1)
        uint32_t w[61] = {0};
        for (int p = 0; p < 31; p++)
            if (a[p]) for (int q = 0; q < 31; q++) w[p + q] += a[p] * b[q];
        for (int d = 60; d >= 31; d--) { w[d - 3] += w[d]; w[d - 31] += w[d]; }
        for (int p = 0; p < 31; p++) out[p] = w[p];
    }
    void jump(uint64_t n) {                    // as if next_fast() had been called n times
        if (n == 0) return;
        uint32_t y[65];                        // the current history o[-34] .. o[-1], oldest first, and 31 words beyond it
        for (int k = 0; k < 34; k++) { int j = i + k; y[k] = r[j >= 34 ? j - 34 : j]; }
        for (int k = 34; k < 65; k++) y[k] = y[k - 3] + y[k - 31];
        uint32_t acc[31] = {1}, base[31] = {0, 1}, tmp[31];      // acc = 1, base = t
        for (uint64_t e = n; e; e >>= 1) {
            if (e & 1) { poly_mul(acc, base, tmp); memcpy(acc, tmp, sizeof tmp); }
            if (e >> 1) { poly_mul(base, base, tmp); memcpy(base, tmp, sizeof tmp); }
        }
        for (int j = 0; j < 34; j++) {         // y[n + j] = sum_c acc[c] * y[c + j]
            uint32_t v = 0;
            for (int c = 0; c < 31; c++) v += acc[c] * y[c + j];
            r[j] = v;
        }
        i = 0;
    }
    // Is libc's rand() this generator?  Decided once per process under a lock: 64 sequential draws from srand(42), then
    // the stream entered at draw 100 000 by jump() against libc's own 100 000th..100 063rd draws.  Leaves libc freshly
    // seeded with 42 the one time it runs.
    static bool libc_is_this_generator() {
        static std::once_flag once;
        static bool same = false;
        std::call_once(once, [] {
            GlibcRandStream g, j;
            g.seed(42u);
            j.seed(42u);
            j.jump(100000);
            srand(42u);
            bool ok = true;
            for (int k = 0; k < 64; k++) ok &= rand() == g.next_fast();
            for (int k = 64; k < 100000; k++) (void)rand();
            for (int k = 0; k < 64; k++) ok &= rand() == j.next_fast();
            srand(42u);
            same = ok;
        });
        return same;
    }
    // the stream as srand(sd) leaves it; `fast` = the restatement may stand in for libc (else next() draws from rand(),
    // which the caller has seeded)
    void start(uint32_t sd = 42u) {
        fast = libc_is_this_generator();
        seed(sd);
    }
};

}  // namespace bamm
