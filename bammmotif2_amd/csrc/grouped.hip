// Host side of the grouped-column EM kernel (k_em_grp, grouped_kernel.h has the kernel and its design notes):
// which length classes it is built for, the LDS geometry of its tables, the planner that picks group size,
// private copies and table layout for a (K, W), and the dispatch of the 2..16 positions-per-lane classes
// (20..32 live in grouped_long.hip, 40 / 48 in grouped_xl.hip: translation units of their own, compiled side by side).

#include "grouped_kernel.h"

namespace bamm {

// length classes the grouped kernel is instantiated for: 2..48 positions per lane (a plan needs M >= G)
bool grp_supported_class(int M, uint32_t K) { return K <= 3u && M >= 2 && M <= 64; }
// 4 / 3 / 2 / 1 waves per SIMD: the grouped kernel's register budget runs out one class earlier than k_em_seq's
uint32_t grp_max_threads(int M) { return M <= 8 ? 1024u : (M <= 16 ? 768u : (M <= 32 ? 512u : 256u)); }

// layout: bit 0 = groups cut by the LW1 edge as per-wave virtual rows (else partial table rows),
//         bit 1 = odd number of quads per odds-table row,
//         bit 2 = odd number of cells per count-table row: with an even number of groups T the rows of the
//                 count table start on every 2nd / 4th / 8th bank pair only (T = 10: 80-byte rows reach half of
//                 the banks, T = 8: a quarter) and the ds_add_u64 of a wave pile up there; one spare cell per row
//                 spreads them over all banks (K = 3, W = 20: 1.88 -> see DESIGN.md)
// Mixed rows (mixed_kernel.h): K = 2, W = 3 B + 4 A with A = W mod 3 in {1, 2} wide groups.  Row counts are
// compile-time constants of the kernel (16 or 12 waves per block, 9 virtual rows per wave).
bool mix_geometry(uint32_t K, uint32_t W, int M, uint32_t waves, bool accum, GrpGeom* out) {
    if (K != 2u || M < 4 || 2 * (M - 1) + 12 > 32 || (waves != 16u && waves != 12u)) return false;
    if (waves != grp_max_threads(M) / 64u) return false;
    const uint32_t A = W % 3u;
    if (A == 0u || W < 4u * A + 3u) return false;
    const uint32_t B = (W - 4u * A) / 3u, T = A + B, Bj = 6u, Bv = 9u;
    if (B > 8u || Bv * T > 64u) return false;
    // measured (tools/mix_probe.py, 1M x 200 bp): the group saved pays when the narrow groups fill one quad and the
    // motif is not short (W = 13, 14, 16, 17, 20: 3-5 % faster than the uniform rows; W = 10, 11, 19, 22: slower)
    if (B > 4u || W < 13u) return false;
    GrpGeom g{};
    g.G = 4u; g.T = T; g.Tq = (B + 3u) / 4u; g.delta = 0u; g.Ts = 0u; g.Rf = 1024u;
    g.np = 0u; g.Rn = 1024u; g.R0 = 1025u; g.Bj = Bj; g.Bv = Bv; g.Rtot = 1025u + waves * Bv;
    g.mixA = A; g.mixB = B; g.cap = 1u;
    const uint32_t R6T = 4097u + waves * Bv, Y = 64u;
    auto up16 = [](uint32_t x) { return (x + 15u) & ~15u; };
    for (int odd = 1; odd >= 0; odd--) {                     // narrow odds rows of an odd number of quads when that fits
        g.rowstride = (odd ? (g.Tq | 1u) : g.Tq) * 4u;
        g.layout = 8u | 1u | (odd ? 2u : 0u);
        uint32_t off = 0;
        g.off_sg6 = off; off = up16(off + R6T * A * 4u);
        g.off_sg = off; off = up16(off + g.Rtot * g.rowstride * 4u);
        g.off_stat = g.off_sg;                               // epilogue only: over the narrow odds
        g.off_n1 = g.off_sg6;                                // epilogue only: over the wide odds
        if (16u * 3u * 8u > g.Rtot * g.rowstride * 4u || (accum && W * Y * 8u > R6T * A * 4u)) continue;
        // the single-column table is staged for the prologue where the counts will be (E-only launches: behind
        // the odds); the fix lanes read it from global memory
        g.off_ng6 = off;
        g.off_s1 = off;
        const uint32_t counts = accum ? up16(A * R6T * 8u) + up16(B * g.Rtot * 8u) : 0u;
        g.off_ng = off + (accum ? up16(A * R6T * 8u) : 0u);
        off += std::max(counts, up16(W * (Y + 1u) * 4u));
        // resident bins for as many of the motif's leading columns as fit (the fix lanes log the rest)
        g.off_wave = off;
        g.wave_bytes = 0u;
        if (accum && off <= 160u * 1024u) {
            const uint32_t cols = std::min(W, (160u * 1024u - off) / (Y * 8u));
            g.wave_bytes = cols * Y * 8u;
            off += g.wave_bytes;
        }
        g.cap = 0u;                                          // the single-column table is not resident
        g.lds_bytes = off;
        if (off <= 160u * 1024u) { *out = g; return true; }
        if (g.Tq & 1u) break;                                // the stride was odd already
    }
    return false;
}

bool grp_geometry(uint32_t K, uint32_t W, uint32_t G, int M, uint32_t waves, bool accum, uint32_t logC, uint32_t layout,
                  GrpGeom* out) {
    if (layout & 8u) return G == 4u && logC == 0u && mix_geometry(K, W, M, waves, accum, out);
    if (K > 3u || W == 0u || G < 2u || G > 4u || K + G > 5u || (int)G > M) return false;
    const bool fixg = K == 3u;                               // single-column table in global memory unless it fits, bins in the epilogue
    GrpGeom g{};
    g.G = G;
    g.T = (W + g.G - 1u) / g.G;
    if (g.T > 64u) return false;
    g.Tq = (g.T + 3u) / 4u;
    if (M > 48 && g.Tq > 4u) return false;                  // 56 / 64 positions per lane: the straight-line chain only
    g.delta = g.G * g.T - W;
    g.Ts = (layout & 4u) ? (g.T | 1u) : g.T;
    g.Rf = 1u << (2u * (K + g.G));
    g.layout = layout;
    g.np = (layout & 1u) ? 0u : g.G - 1u;
    uint32_t r = g.Rf;
    for (uint32_t d = 0; d < g.np; d++) {                   // partial class d: d+1 trailing positions neutral
        g.base[d] = r;
        g.psize[d] = 1u << (2u * (K + g.G - 1u - d));
        r += g.psize[d];
    }
    g.Rn = r;                                                // neutral row
    g.R0 = r + 1u;
    // virtual rows per wave: Bj for the group ends next to N exceptions (one N makes K exceptions in a
    // row, i.e. K-1+G group ends; two more rows cover "NN" and "N.N"), G-1 more for the group ends cut
    // by the LW1 edge when the table has no partial rows for them; one fix lane per (row, group)
    const uint32_t edge = (layout & 1u) ? g.G - 1u : 0u;
    const uint32_t Y = 1u << (2u * (K + 1u));
    auto up16 = [](uint32_t x) { return (x + 15u) & ~15u; };
    const uint32_t kLds = 160u * 1024u;
    // everything that depends on the number of virtual rows per wave; s1_lds: the single-column table in LDS
    auto lay_out = [&](uint32_t Bj, bool s1_lds) {
        g.Bj = Bj;
        g.Bv = g.Bj + edge;
        g.Rtot = g.R0 + waves * g.Bv;
        uint32_t off = 0;
        // layout bit 1, an odd number of quads per row: rows then start on all 16 bank-quads, not on every 2nd / 4th
        // one, and the 16 lanes of a ds_read_b128 group that read the same quad index of random rows spread over
        // all of them (with 2 quads per row a group shared 8)
        g.rowstride = ((layout & 2u) ? (g.Tq | 1u) : g.Tq) * 4u;
        g.off_sg = off; off = up16(off + g.rowstride * g.Rtot * 4u);
        g.off_s1 = off; if (s1_lds) off = up16(off + W * (Y + 1u) * 4u);
        g.off_stat = off; off = up16(off + 16u * 3u * 8u);
        g.off_ng = off;
        if (accum) off = up16(off + ((g.Ts * g.Rtot) << logC) * 8u);
        g.off_n1 = off;
        if (accum && !fixg) off = up16(off + W * Y * 8u);
        // K = 3: no room beside the tables; the bins take the odds table's place once the block's sequences are done
        if (fixg) {
            g.off_n1 = g.off_sg;
            if (accum && W * Y * 8u > g.rowstride * g.Rtot * 4u) return false;
        }
        g.off_wave = off;
        g.cap = s1_lds ? 1u : 0u;                            // K = 3 kernels: where the single-column table lives
        g.wave_bytes = 0u;
        g.lds_bytes = off;
        return off <= kLds;
    };
    uint32_t Bj = std::min(8u, K + g.G + 1u);
    while (Bj > 0u && (Bj + edge) * g.T > 64u) Bj--;
    if ((Bj + edge) * g.T > 64u) return false;
    bool ok;
    if (!fixg) {
        ok = lay_out(Bj, true);
    } else {
        // K = 3: the single-column table (fix lanes only) stays in LDS when it fits -- a fix lane that has to
        // fetch its odds through L2 stalls its wave once per sequence -- if need be with one virtual row per
        // wave fewer (K-1+G rows cover one N; "NN" / "N.N" then go to the per-column kernel); else global memory
        const uint32_t Bmin = std::min(Bj, K - 1u + g.G);
        ok = lay_out(Bj, true) || lay_out(Bmin, true) || lay_out(Bj, false);
    }
    *out = g;
    return ok;
}

// group size and private copies for (K, W): wider groups first (fewer gathers / adds per window),
// then as many private copies of the count table as the 160 KiB hold
// `many_exceptions`: most sequences of the launch carry exceptions anyway (double-stranded sets: the
// strand junction), so the fix lanes run per sequence whether or not the edge rows are virtual.
// forced / forced_layout: bamm_ctx_set_tuning("group_size" / "group_layout"), 0 / -1 = the planner's choice
bool grp_plan(uint32_t K, uint32_t W, int M, uint32_t waves, bool many_exceptions, bool enough_work, uint32_t forced,
              int forced_layout, uint32_t* G_out, uint32_t* logC_out, uint32_t* layout_out) {
    // layouts in order of preference.  Sets with exceptions everywhere run the fix lanes per sequence
    // anyway: virtual edge rows cost them nothing extra, save the decode's partial-row patches and make
    // the tables small enough for the odd stride (K=2: 1.08 -> 1.01 ms, K=1: 1.05 -> 0.98 ms).  Clean
    // sets (single strand, K=0) would pay the fix lanes for the edge alone (+11 %): partial rows there.
    // each with the count rows padded to an odd number of cells first (bit 2; the same thing when T is odd)
    const uint32_t order_exc[6] = {7u, 3u, 6u, 2u, 4u, 0u}, order_clean[6] = {6u, 2u, 4u, 0u, 7u, 3u};
    // K = 2, W not a multiple of 3: one group less with the motif's last columns on 6-mer rows (mixed_kernel.h);
    // its edge rows are virtual, so like layouts 3 / 7 it is for sets that run the fix lanes per sequence anyway
    if (K == 2u && many_exceptions && (forced == 0u || forced == 4u) && ((forced_layout < 0 && enough_work) || forced_layout == 8)) {
        GrpGeom g;
        if (mix_geometry(K, W, M, waves, true, &g)) {
            *G_out = 4u; *logC_out = 0u; *layout_out = g.layout;
            return true;
        }
    }
    if (forced_layout == 8) return false;
    for (uint32_t G = 5u - K; G >= 2u && G + 1u >= 5u - K; G--) {          // G = 5-K, then 4-K
        if (G > 4u || (forced && G != forced)) continue;
        for (int li = 0; li < 6; li++) {
            const uint32_t layout = many_exceptions ? order_exc[li] : order_clean[li];
            if (forced_layout >= 0 && (uint32_t)forced_layout != (layout & 3u)) continue;   // the tuning switch names bits 0-1
            for (int lc = 2; lc >= 0; lc--) {
                GrpGeom g;
                if (grp_geometry(K, W, G, M, waves, true, (uint32_t)lc, layout, &g)) {
                    *G_out = G; *logC_out = (uint32_t)lc; *layout_out = layout;
                    return true;
                }
            }
        }
    }
    return false;
}

// 2..16 positions per lane are instantiated here, 20..32 in grouped_long.hip and 40 / 48 in grouped_xl.hip
// (translation units of their own: the three parts compile side by side)
#define BAMM_FOR_EACH_GCLASS(X) \
    X(1, 2, 1024) X(2, 3, 1024) X(3, 4, 1024) X(4, 5, 1024) X(5, 6, 1024) X(6, 7, 1024) X(7, 8, 1024) X(8, 10, 768) X(9, 12, 768) X(10, 14, 768) X(11, 16, 768)

int launch_em_grp(int mclass, bool accum, bool write_r, const GrpKernelArgs& a, uint32_t blocks, uint32_t threads,
                  hipStream_t st) {
    if (a.g.lds_bytes > 160u * 1024u || blocks == 0 || (threads & 63u) || threads == 0 ||
        threads > grp_max_threads(kMClasses[mclass])) {
        set_error("bad launch of the grouped kernel (%u x %u, %u bytes of LDS)", blocks, threads, a.g.lds_bytes);
        return BAMM_ERR_ARG;
    }
    if (a.g.layout & 8u)
        return a.g.mixA == 2u ? launch_em_mix(mclass, accum, write_r, a, blocks, threads, st)
                              : launch_em_mix1(mclass, accum, write_r, a, blocks, threads, st);
    if (kMClasses[mclass] > 16) return launch_em_grp_long(mclass, accum, write_r, a, blocks, threads, st);
    const uint32_t KG = a.e.K + a.g.G;                       // row = (K+G)-mer: 4 or 5 bases
    switch (mclass * 64 + (int)a.g.G * 8 + (int)KG) {
        BAMM_FOR_EACH_GCLASS(BAMM_GRP_CASES)
        default:
            set_error("no grouped kernel for M class %d, G=%u", mclass, a.g.G);
            return BAMM_ERR_UNSUPPORTED;
    }
    BAMM_HIP(hipGetLastError());
    return BAMM_OK;
}

}  // namespace bamm
