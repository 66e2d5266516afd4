"""The grouped-column kernel (bammmotif2_amd/csrc/grouped.hip) against the pinned oracle, and against
the one-column-at-a-time kernel it replaces for orders K <= 3.

What is specific to it and therefore tested here: G = 5-K or 4-K columns per table row (K = 0..3), the
partial rows at the EM.cpp:167 truncation edge (W not a multiple of G, W < G), the virtual rows next
to N exceptions (Sequence.cpp:38; the strand junction of every double-stranded sequence, and N inside
a sequence), the split of a bucket into sequences the kernel takes and the rest, and every length
class it is built for (4..16 positions per lane).
"""
import os

import numpy as np
import pytest

import bammmotif2_amd as bm
from tests.cases import Case

pytestmark = pytest.mark.gpu

GROUPED_CASES = [
    dict(name="g_k2_ds_m5", N=80, L0=150, W=9, K=2, n_frac=0.0, ragged=20),
    dict(name="g_k2_ds_m7_N", N=120, L0=200, W=20, K=2, n_frac=0.004, ragged=0),
    dict(name="g_k1_ds_N", N=64, L0=200, W=7, K=1, n_frac=0.01, ragged=30),
    dict(name="g_k0_ss", N=64, L0=300, W=13, K=0, ss=True, n_frac=0.005, ragged=40),
    dict(name="g_k0_ds_w2", N=48, L0=170, W=2, K=0, ragged=10),
    dict(name="g_k2_w1", N=40, L0=160, W=1, K=2, ragged=12),
    dict(name="g_k2_w3", N=40, L0=160, W=3, K=2, n_frac=0.003, ragged=12),
    dict(name="g_k2_m10", N=40, L0=300, W=20, K=2, ragged=0),
    dict(name="g_k2_m16", N=24, L0=480, W=21, K=2, n_frac=0.001, ragged=60),
    dict(name="g_k1_ss_m4", N=64, L0=250, W=12, K=1, ss=True, ragged=30),
    dict(name="g_k2_wide", N=32, L0=200, W=31, K=2, ragged=0),
    # the remaining length classes (6, 8, 12, 14 positions per lane)
    dict(name="g_k2_m6", N=24, L0=180, W=10, K=2, n_frac=0.002, ragged=8),
    dict(name="g_k1_m8", N=24, L0=230, W=9, K=1, ragged=8),
    dict(name="g_k2_m12", N=24, L0=370, W=14, K=2, ragged=10),
    dict(name="g_k0_m14", N=24, L0=430, W=11, K=0, n_frac=0.002, ragged=10),
    # more than 16 groups (the E-chain's loop beyond four quads), single strand so that no sequence
    # needs more virtual rows than 64 / T fix lanes allow
    dict(name="g_k2_w40_ss", N=32, L0=400, W=40, K=2, ss=True, ragged=20),
    dict(name="g_k1_w50_ss", N=32, L0=400, W=50, K=1, ss=True, ragged=20),
    # 20, 24, 28 and 32 positions per lane (512-thread blocks): ds sequences of 513..1023 bp
    dict(name="g_k2_m20_24", N=20, L0=640, W=20, K=2, n_frac=0.001, ragged=110),      # L 1061..1501
    dict(name="g_k2_m28_32", N=16, L0=890, W=17, K=2, n_frac=0.001, ragged=125),      # L 1531..2031
    dict(name="g_k1_m32_ss", N=16, L0=1900, W=12, K=1, ss=True, ragged=140),
    dict(name="g_k0_m24_ss", N=16, L0=1400, W=9, K=0, ss=True, n_frac=0.002, ragged=120),
    # 40 and 48 positions per lane (256-thread blocks): 2049..3072 positions
    dict(name="g_k2_m40_48", N=10, L0=1280, W=20, K=2, n_frac=0.0005, ragged=240),    # L 2081..3041
    dict(name="g_k1_m40_48_ss", N=10, L0=2600, W=14, K=1, ss=True, n_frac=0.0005, ragged=460),
    dict(name="g_k0_m48", N=6, L0=1500, W=8, K=0, ragged=30),                        # L 2941..3061
    # 56 and 64 positions per lane: 3073..4096 positions (the E-chain reads one slot at a time there)
    dict(name="g_k2_m56_64", N=8, L0=1600, W=20, K=2, n_frac=0.0003, ragged=440),     # L 3201..4081
    dict(name="g_k1_m64_ss", N=6, L0=3900, W=14, K=1, ss=True, ragged=190),           # L 3900..4090
    dict(name="g_k3_m56", N=6, L0=1560, W=12, K=3, n_frac=0.0003, ragged=200),        # L 3121..3521
    # mixed rows (csrc/mixed_kernel.h: K = 2, both strands, W = 3 B + 4 A): one and two wide groups, 4..8 positions per lane
    dict(name="g_mix_a1_m4", N=96, L0=110, W=13, K=2, ragged=10),
    dict(name="g_mix_a2_m5", N=96, L0=140, W=14, K=2, n_frac=0.002, ragged=12),
    dict(name="g_mix_a1_m6", N=80, L0=170, W=16, K=2, n_frac=0.002, ragged=14),
    dict(name="g_mix_a2_m8", N=64, L0=235, W=17, K=2, n_frac=0.002, ragged=16),
    # 2 and 3 positions per lane (short reads: 65..192 positions), where a group is as wide as a lane's run
    dict(name="g_k2_m2_ds", N=200, L0=40, W=12, K=2, n_frac=0.002, ragged=6),        # L 69..93: G = 2
    dict(name="g_k2_m3_ds", N=160, L0=80, W=10, K=2, n_frac=0.002, ragged=12),       # L 137..185: G = 3
    dict(name="g_k1_m3_ss", N=160, L0=160, W=8, K=1, ss=True, ragged=30),            # G = 3 (4 does not fit a lane)
    # K = 3: two columns per 5-mer row, single-column table and virtual-row bins in global memory, 10-bit record fields
    dict(name="g_k3_ds_m7", N=120, L0=200, W=20, K=3, n_frac=0.002, ragged=0),       # the bench shape at k = 3
    dict(name="g_k3_ds_m5_odd", N=80, L0=150, W=13, K=3, n_frac=0.004, ragged=20),   # W odd: a group cut in front
    dict(name="g_k3_ss", N=64, L0=300, W=10, K=3, ss=True, n_frac=0.005, ragged=40),
    dict(name="g_k3_w1", N=40, L0=160, W=1, K=3, ragged=12),
    dict(name="g_k3_w22", N=32, L0=260, W=22, K=3, ragged=10),                       # the widest motif whose tables fit
    dict(name="g_k3_m16_24", N=20, L0=560, W=16, K=3, n_frac=0.001, ragged=100),     # L 921..1121
    dict(name="g_k3_m40_48", N=8, L0=1400, W=12, K=3, n_frac=0.0005, ragged=200),    # L 2401..2801
    dict(name="g_k3_m2_ds", N=200, L0=40, W=8, K=3, n_frac=0.002, ragged=6),
]


def make_em(ctx, c, orc, **kw):
    seq, kmer, off, vbg = c.encode(orc)
    pk = bm.PackedSeqs.from_kmers(kmer, off)
    ss = bm.SeqSet(ctx, pk)
    em = bm.EM(ctx, ss, c.K, c.W, vbg, c.A, c.v0, c.q, bg_order=c.bg_order, **kw)
    return em, ss, kmer, off, vbg


@pytest.mark.parametrize("spec", GROUPED_CASES, ids=[d["name"] for d in GROUPED_CASES])
def test_grouped_kernel_matches_oracle(spec, gpu_ctx, orc):
    c = Case(**spec)
    em, ss, kmer, off, vbg = make_em(gpu_ctx, c, orc)
    grouped, other, _ = em.plan()
    assert grouped > 0, "this case is meant to exercise the grouped kernel"
    if c.n_frac >= 0.004 and not c.ss:
        assert other > 0, "sequences with scattered N must fall to the per-column kernel"
    Kb = min(c.bg_order, c.K)
    # beyond ~2500 windows the reference's sequential fp32 Z sum (EM.cpp:179-182) is itself 1e-5 off, and
    # every r of the sequence carries that factor (test_grouped_kernel_matches_exact_arithmetic has the
    # tight bar for these lengths)
    long_seq = max(1.0, 4e-4 * c.L0 * (1 if c.ss else 2))
    for it in range(3):
        v = em.getV()
        em.EStep()
        s_o = orc.linear_s(v, vbg, c.K, c.W, Kb)
        r_o, llh_o = orc.estep(kmer, off, c.K, c.W, s_o, c.q)
        r_g = em.getR()
        # products are rounded group-wise, not left to right: a few 2^-24 per window
        np.testing.assert_allclose(r_g, r_o, rtol=1e-5 * long_seq, atol=1e-12)
        assert np.array_equal(r_g == 0, r_o == 0)
        # Z_n is a wave tree sum here and a sequential fp32 loop over L-W+1 near-equal terms in the
        # reference (EM.cpp:179-182), whose rounding is one-sided when the terms are alike (W <= 2
        # seeds): up to ~1e-6 absolute per sequence on log Z_n
        np.testing.assert_allclose(em.getLLH(), llh_o, rtol=2e-6, atol=2e-6 * c.N * long_seq ** 2)
        em.MStep()
        n_o = orc.mstep_counts(kmer, off, c.K, c.W, r_o)
        # the oracle accumulates N*(L-W+1) fp32 addends per cell like the reference; with
        # W <= 2 every window carries weight, which is its worst case (~sqrt(n) * 2^-24)
        np.testing.assert_allclose(em.getCounts(), n_o, rtol=(3e-5 if c.W <= 2 else 1e-5) * long_seq, atol=1e-6)
        np.testing.assert_allclose(em.getV(), orc.update_v(n_o, c.A, vbg, c.K, c.W),
                                   rtol=(3e-5 if c.W <= 2 else 1e-5) * long_seq, atol=1e-9)
    em.close(); ss.close()


@pytest.mark.parametrize("spec", GROUPED_CASES, ids=[d["name"] for d in GROUPED_CASES])
def test_grouped_kernel_matches_exact_arithmetic(spec, gpu_ctx, orc):
    """One E+M step against the oracle's fp64 restatement of the same formulas: the fixed-point
    counts carry no accumulation noise, so this bar is ten times tighter than the parity bar."""
    c = Case(**spec)
    em, ss, kmer, off, vbg = make_em(gpu_ctx, c, orc)
    v64, *_ = orc.em_step_f64(kmer, off, c.K, c.W, c.bg_order, vbg, c.A, c.v0, c.q)
    em.iterate(1)
    np.testing.assert_allclose(em.getV(), v64, rtol=1e-6, atol=1e-9)
    em.close(); ss.close()


_VS_PER_COLUMN = GROUPED_CASES[:5] + [d for d in GROUPED_CASES if d["name"] in ("g_k3_ds_m7", "g_k3_ds_m5_odd")]


@pytest.mark.parametrize("spec", _VS_PER_COLUMN, ids=[d["name"] for d in _VS_PER_COLUMN])
def test_grouped_equals_per_column_kernel(spec, gpu_ctx, orc):
    """Same handle twice, once with the grouped kernel switched off: counts are sums of the same
    fixed-point addends up to the rounding of r, llh and q chain identically."""
    c = Case(**spec)
    res = []
    for off_switch in (False, True):
        gpu_ctx.set_tuning(grouped=not off_switch)
        try:
            em, ss, kmer, off, vbg = make_em(gpu_ctx, c, orc, optimizeQ=True)
        finally:
            gpu_ctx.set_tuning(grouped=1)
        g, o, _ = em.plan()
        assert (g == 0) == off_switch
        em.iterate(4)
        res.append((em.getV(), em.getCounts(), em.getQ(), em.trace()[0], em.getR()))
        em.close(); ss.close()
    a, b = res
    np.testing.assert_allclose(a[0], b[0], rtol=2e-6, atol=1e-10)
    np.testing.assert_allclose(a[1], b[1], rtol=2e-6, atol=1e-6)
    np.testing.assert_allclose(a[2], b[2], rtol=1e-6)
    np.testing.assert_allclose(a[3], b[3], rtol=1e-5, atol=5e-7 * c.N)     # log Z_n of a Z_n that differs by 1 ulp
    np.testing.assert_allclose(a[4], b[4], rtol=5e-6, atol=1e-12)


def test_group_sizes_agree(gpu_ctx, orc):
    """K = 2 runs with 3 columns per row when the tables fit and with 2 otherwise: the two only differ
    in how the window products are rounded."""
    c = Case(name="g_sizes", N=300, L0=200, W=20, K=2)
    out = []
    for G in (3, 2):
        gpu_ctx.set_tuning(group_size=G)
        try:
            em, ss, *_ = make_em(gpu_ctx, c, orc)
        finally:
            gpu_ctx.set_tuning(group_size=0)
        assert em.plan()[0] == c.N
        em.iterate(6)
        out.append((em.getCounts(), em.getV(), em.trace()[0]))
        em.close(); ss.close()
    np.testing.assert_allclose(out[0][0], out[1][0], rtol=2e-6, atol=1e-6)
    np.testing.assert_allclose(out[0][1], out[1][1], rtol=2e-6, atol=1e-10)
    np.testing.assert_allclose(out[0][2], out[1][2], rtol=1e-6)


@pytest.mark.parametrize("K,W,layout", [(2, 12, -1), (3, 12, -1), (2, 14, 8)], ids=["k2", "k3", "k2_mixed_rows"])
def test_grouped_with_fold_mask_and_optimize(K, W, layout, gpu_ctx, orc):
    """CV-fold mask + the full optimize() loop through the grouped kernels vs the oracle (K = 3 and the mixed
    rows: sequences the mask skips leave no entry in the fix lanes' log)."""
    c = Case(name="g_opt", N=400, L0=200, W=W, K=K, n_frac=0.0)
    seq, kmer, off, vbg = c.encode(orc)
    pk = bm.PackedSeqs.from_kmers(kmer, off)
    ss = bm.SeqSet(gpu_ctx, pk)
    mask = (np.arange(c.N) % 5 != 2).astype(np.uint8)
    gpu_ctx.set_tuning(group_layout=layout)
    try:
        em = bm.EM(gpu_ctx, ss, c.K, c.W, vbg, c.A, c.v0, c.q, bg_order=c.bg_order, mask=mask, max_iterations=8)
    finally:
        gpu_ctx.set_tuning(group_layout=-1)
    assert em.plan()[0] == c.N and (em.plan_mixed() == c.N) == (layout == 8)
    em.optimize()
    keep = np.flatnonzero(mask)
    sub_off = np.zeros(len(keep) + 1, np.uint64)
    parts = []
    for i, n in enumerate(keep):
        parts.append(kmer[int(off[n]):int(off[n + 1])])
        sub_off[i + 1] = sub_off[i] + np.uint64(len(parts[-1]))
    res = orc.optimize(np.concatenate(parts), sub_off, c.K, c.W, c.bg_order, vbg, c.A, c.v0, c.q, max_iter=8)
    assert em.iteration() == res["iterations"]
    np.testing.assert_allclose(em.getV(), res["v"], rtol=5e-5, atol=1e-8)
    em.close(); ss.close()


def test_handles_created_from_several_host_threads(gpu_ctx, orc):
    """FDR::evaluateMotif builds its EM objects on one sequence set from up to four host threads
    (FDR.cpp:37): handle creation (which builds the per-order tables of the shared set lazily) and the
    runs themselves must be safe to issue concurrently."""
    import threading
    c = Case(name="g_thr", N=500, L0=200, W=12, K=2)
    seq, kmer, off, vbg = c.encode(orc)
    pk = bm.PackedSeqs.from_kmers(kmer, off)
    ss = bm.SeqSet(gpu_ctx, pk)
    folds = 4
    out = [None] * folds

    def run(f):
        mask = (np.arange(c.N) % folds != f).astype(np.uint8)
        em = bm.EM(gpu_ctx, ss, c.K, c.W, vbg, c.A, c.v0, c.q, bg_order=c.bg_order, mask=mask)
        em.iterate(5)
        out[f] = em.getV()
        em.close()

    th = [threading.Thread(target=run, args=(f,)) for f in range(folds)]
    for t in th: t.start()
    for t in th: t.join()
    for f in range(folds):                                   # the same folds one after the other
        mask = (np.arange(c.N) % folds != f).astype(np.uint8)
        em = bm.EM(gpu_ctx, ss, c.K, c.W, vbg, c.A, c.v0, c.q, bg_order=c.bg_order, mask=mask)
        em.iterate(5)
        assert np.array_equal(em.getV(), out[f])
        em.close()
    ss.close()


_MIXED = [d for d in GROUPED_CASES if d["name"].startswith("g_mix_") or d["name"] in ("g_k2_ds_m7_N", "g_k2_m10")]


@pytest.mark.parametrize("spec", _MIXED, ids=[d["name"] for d in _MIXED])
def test_mixed_rows_agree_with_exact_arithmetic_and_uniform_rows(spec, gpu_ctx, orc):
    """k_em_mix (the motif's last W mod 3 groups on 6-mer rows; group_layout 8 -- the planner picks it by itself
    only for launches of tens of thousands of sequences) against the fp64 restatement, the oracle's r, and the
    uniform 5-mer rows (group_layout 3): products are rounded group-wise either way."""
    c = Case(**spec)
    gpu_ctx.set_tuning(group_layout=8)
    try:
        em, ss, kmer, off, vbg = make_em(gpu_ctx, c, orc)
    finally:
        gpu_ctx.set_tuning(group_layout=-1)
    grouped = em.plan()[0]
    assert grouped > 0 and em.plan_mixed() == grouped
    # E-step against the oracle, then one step against exact arithmetic
    em.EStep()
    Kb = min(c.bg_order, c.K)
    r_o, llh_o = orc.estep(kmer, off, c.K, c.W, orc.linear_s(c.v0, vbg, c.K, c.W, Kb), c.q)
    r_g = em.getR()
    np.testing.assert_allclose(r_g, r_o, rtol=1e-5, atol=1e-12)
    assert np.array_equal(r_g == 0, r_o == 0)
    np.testing.assert_allclose(em.getLLH(), llh_o, rtol=2e-6, atol=2e-6 * c.N)
    v64, *_ = orc.em_step_f64(kmer, off, c.K, c.W, c.bg_order, vbg, c.A, c.v0, c.q)
    em.iterate(1)
    np.testing.assert_allclose(em.getV(), v64, rtol=1e-6, atol=1e-9)
    em.iterate(3)
    v_mix, llh_mix, n_mix = em.getV(), em.trace()[0].copy(), em.getCounts()
    em.close()
    gpu_ctx.set_tuning(group_layout=3)
    try:
        em = bm.EM(gpu_ctx, ss, c.K, c.W, vbg, c.A, c.v0, c.q, bg_order=c.bg_order)
    finally:
        gpu_ctx.set_tuning(group_layout=-1)
    # (with scattered N the split differs by a sequence or two: a group of four reaches one position further)
    assert abs(em.plan()[0] - grouped) <= 2 and em.plan_mixed() == 0
    em.iterate(4)
    np.testing.assert_allclose(llh_mix, em.trace()[0], rtol=2e-6)
    np.testing.assert_allclose(n_mix, em.getCounts(), rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(v_mix, em.getV(), rtol=2e-5, atol=1e-9)
    em.close(); ss.close()


def test_planner_picks_mixed_rows_for_large_launches(gpu_ctx, orc):
    """The bench shape (W = 20, K = 2, both strands) with enough sequences goes through k_em_mix by itself; a small
    set of the same shape stays on the uniform rows (the larger tables cost 4-8 us per launch)."""
    for n, mixed in ((60000, True), (3000, False)):
        c = Case(name="g_plan", N=n, L0=200, W=20, K=2, ragged=0)
        em, ss, *_ = make_em(gpu_ctx, c, orc)
        assert em.plan()[0] == n
        assert (em.plan_mixed() == n) == mixed
        em.iterate(2)
        assert np.isfinite(em.getLLH())
        em.close(); ss.close()


@pytest.mark.parametrize("layout", [0, 2, 3])
@pytest.mark.parametrize("spec", [GROUPED_CASES[1], GROUPED_CASES[9], GROUPED_CASES[4]],
                         ids=["k2_ds_N", "k1_ss", "k0_ds_w2"])
def test_every_table_layout_matches_exact_arithmetic(spec, layout, gpu_ctx, orc):
    """The table layouts the planner chooses between (grp_geometry: partial rows or per-wave virtual rows
    for the groups cut by the LW1 edge, even or odd number of quads per row) compute the same thing."""
    c = Case(**spec)
    gpu_ctx.set_tuning(group_layout=layout)
    try:
        em, ss, kmer, off, vbg = make_em(gpu_ctx, c, orc)
    finally:
        gpu_ctx.set_tuning(group_layout=-1)
    assert em.plan()[0] > 0
    v64, *_ = orc.em_step_f64(kmer, off, c.K, c.W, c.bg_order, vbg, c.A, c.v0, c.q)
    em.iterate(1)
    np.testing.assert_allclose(em.getV(), v64, rtol=1e-6, atol=1e-9)
    em.close(); ss.close()
