"""Parity of the HIP path (through the C ABI) against the pinned CPU oracle on seeded inputs.

Tolerances: the window products and the odds / log-odds tables are bit-exact by construction
(same fp32 operation order, IEEE division); the per-sequence partition sum is a wave tree sum
instead of the reference's sequential fp32 loop and the counts are accumulated per block and
reduced in fp64, so r / llh / n / v carry summation-order noise only.  The bar BASELINE.json
states is 1e-5 relative on the learned conditional probabilities.
"""
import numpy as np
import pytest

import bammmotif2_amd as bm
from tests.cases import SMALL_CASES, Case

pytestmark = pytest.mark.gpu

V_RTOL = 1e-5          # north_star: conditional probabilities within 1e-5 relative
R_RTOL, R_ATOL = 1e-5, 1e-12   # the oracle's sequential fp32 sum over L-W+1 terms carries ~sqrt(L)*6e-8
LLH_RTOL = 2e-6


def make_em(ctx, c, orc, **kw):
    seq, kmer, off, vbg = c.encode(orc)
    pk = bm.PackedSeqs.from_kmers(kmer, off)
    ss = bm.SeqSet(ctx, pk)
    em = bm.EM(ctx, ss, c.K, c.W, vbg, c.A, c.v0, c.q, bg_order=c.bg_order, **kw)
    return em, ss, kmer, off, vbg


@pytest.mark.parametrize("spec", SMALL_CASES, ids=[d["name"] for d in SMALL_CASES])
def test_estep_mstep_match_oracle(spec, gpu_ctx, orc):
    c = Case(**spec)
    em, ss, kmer, off, vbg = make_em(gpu_ctx, c, orc)
    Kb = min(c.bg_order, c.K)
    q = c.q
    for it in range(3):
        # step-wise parity: every pass starts from the device's own current model, so the
        # oracle sees exactly the inputs the kernels saw
        v = em.getV()
        if it == 0:
            assert np.array_equal(v, c.v0)
        em.EStep()
        s_o = orc.linear_s(v, vbg, c.K, c.W, Kb)
        assert np.array_equal(em.getS(), s_o)                       # Motif.cpp:485-494, bit-exact
        r_o, llh_o = orc.estep(kmer, off, c.K, c.W, s_o, q)
        r_g = em.getR()
        np.testing.assert_allclose(r_g, r_o, rtol=R_RTOL, atol=R_ATOL)
        assert np.array_equal(r_g == 0, r_o == 0)                   # unused slots stay exactly zero
        # log Z_n of a Z_n rounded to fp32 carries ~6e-8 absolute noise per sequence
        np.testing.assert_allclose(em.getLLH(), llh_o, rtol=LLH_RTOL, atol=5e-7 * c.N)
        em.MStep()
        n_o = orc.mstep_counts(kmer, off, c.K, c.W, r_o)
        np.testing.assert_allclose(em.getCounts(), n_o, rtol=1e-5, atol=1e-6)
        v_o = orc.update_v(n_o, c.A, vbg, c.K, c.W)
        np.testing.assert_allclose(em.getV(), v_o, rtol=V_RTOL, atol=1e-9)
        assert em.getQ() == np.float32(q)                           # MStep never touches q
    em.close(); ss.close()


@pytest.mark.parametrize("spec", SMALL_CASES[:4], ids=[d["name"] for d in SMALL_CASES[:4]])
def test_iterate_equals_estep_mstep_and_oracle(spec, gpu_ctx, orc):
    c = Case(**spec)
    em, ss, kmer, off, vbg = make_em(gpu_ctx, c, orc, optimizeQ=True)
    em.iterate(6)
    res = orc.optimize(kmer, off, c.K, c.W, c.bg_order, vbg, c.A, c.v0, c.q, optimizeQ=True,
                       epsilon=0.0, max_iter=6)
    assert res["iterations"] == 6
    np.testing.assert_allclose(em.getV(), res["v"], rtol=5e-5, atol=1e-8)
    np.testing.assert_allclose(em.getQ(), res["q"], rtol=1e-5)
    llh, vd, q = em.trace()
    np.testing.assert_allclose(llh, res["trace_llh"], rtol=1e-5)
    np.testing.assert_allclose(vd, res["trace_vdiff"], rtol=1e-3, atol=1e-6)
    assert em.iteration() == 6
    em.close(); ss.close()


def test_getR_after_iterate_is_the_last_estep(gpu_ctx, orc):
    """EM::write / getR() after optimize() see the r of the last EStep, i.e. computed from the
    model BEFORE the last MStep (EM.cpp:93-96,583-601)."""
    c = Case(**SMALL_CASES[0])
    em, ss, kmer, off, vbg = make_em(gpu_ctx, c, orc, optimizeQ=True)
    em.iterate(1)
    v1, q1 = em.getV(), em.getQ()
    em.iterate(1)
    s1 = orc.linear_s(v1, vbg, c.K, c.W, min(c.bg_order, c.K))
    r_o, _ = orc.estep(kmer, off, c.K, c.W, s1, q1)
    np.testing.assert_allclose(em.getR(), r_o, rtol=R_RTOL, atol=R_ATOL)
    # EStep(); optimize_q(); MStep() : the MStep still uses the r of that EStep
    em.EStep()
    v2, q2 = em.getV(), em.getQ()
    em.optimize_q()
    assert em.getQ() != np.float32(q2)
    em.MStep()
    s2 = orc.linear_s(v2, vbg, c.K, c.W, min(c.bg_order, c.K))
    r2, _ = orc.estep(kmer, off, c.K, c.W, s2, q2)
    v3 = orc.update_v(orc.mstep_counts(kmer, off, c.K, c.W, r2), c.A, vbg, c.K, c.W)
    np.testing.assert_allclose(em.getV(), v3, rtol=V_RTOL, atol=1e-9)
    em.close(); ss.close()


def test_getR_after_mstep_then_optimize_q(gpu_ctx, orc):
    """The reference's own call order (EM.cpp:93-99): EStep(); MStep(); optimize_q().  getR() afterwards is
    still the r of that EStep, computed with the q it saw; the new q only enters the next EStep."""
    c = Case(**SMALL_CASES[0])
    em, ss, kmer, off, vbg = make_em(gpu_ctx, c, orc, optimizeQ=True)
    Kb = min(c.bg_order, c.K)
    v0, q0 = em.getV(), em.getQ()
    em.EStep()
    em.MStep()
    v1 = em.getV()
    em.optimize_q()
    s0 = orc.linear_s(v0, vbg, c.K, c.W, Kb)
    r0, _ = orc.estep(kmer, off, c.K, c.W, s0, q0)
    q1 = orc.optimize_q(r0, off, c.W)
    assert q1 != np.float32(q0)
    np.testing.assert_allclose(em.getQ(), q1, rtol=1e-5)
    np.testing.assert_allclose(em.getR(), r0, rtol=R_RTOL, atol=R_ATOL)          # old s, OLD q
    # ... and the other order, EStep(); optimize_q(); MStep(); getR()
    em.EStep()
    s1 = orc.linear_s(v1, vbg, c.K, c.W, Kb)
    r1, _ = orc.estep(kmer, off, c.K, c.W, s1, q1)
    em.optimize_q()
    em.MStep()
    np.testing.assert_allclose(em.getR(), r1, rtol=R_RTOL, atol=R_ATOL)
    np.testing.assert_allclose(em.getQ(), orc.optimize_q(r1, off, c.W), rtol=1e-5)
    em.close(); ss.close()


def test_second_optimize_call_reestimates_q(gpu_ctx, orc):
    """`iteration` is local to EM::optimize (EM.cpp:75-99): every call re-estimates q in its first five
    passes, whatever the handle ran before."""
    c = Case(**SMALL_CASES[0])
    em, ss, kmer, off, vbg = make_em(gpu_ctx, c, orc, optimizeQ=True, epsilon=0.0, max_iterations=3)
    res1 = orc.optimize(kmer, off, c.K, c.W, c.bg_order, vbg, c.A, c.v0, c.q, optimizeQ=True, epsilon=0.0, max_iter=3)
    res2 = orc.optimize(kmer, off, c.K, c.W, c.bg_order, vbg, c.A, res1["v"], res1["q"], optimizeQ=True, epsilon=0.0,
                        max_iter=3)
    assert em.optimize() == 3
    np.testing.assert_allclose(em.getQ(), res1["q"], rtol=1e-5)
    assert em.optimize() == 3
    assert res2["q"] != res1["q"]
    np.testing.assert_allclose(em.getQ(), res2["q"], rtol=2e-5)
    np.testing.assert_allclose(em.getV(), res2["v"], rtol=5e-5, atol=1e-8)
    em.close(); ss.close()


SLICED_CASES = [
    dict(name="k4", N=60, L0=300, W=30, K=4, ss=True, ragged=40, n_frac=0.01),
    dict(name="k4_ds_config4_shape", N=48, L0=500, W=30, K=4, ragged=12, n_frac=0.002),    # both strands, L ~ 1001: 16 positions per lane, 768 threads
    dict(name="k4_M96_128", N=5, L0=6600, W=30, K=4, ss=True, ragged=1500, n_frac=0.0005),   # longest length classes
    dict(name="k5_w12", N=30, L0=400, W=12, K=5, ragged=60, n_frac=0.002),                  # E slices carry the chain through HBM
]


@pytest.mark.parametrize("spec", SLICED_CASES, ids=[d["name"] for d in SLICED_CASES])
def test_sliced_path_for_large_tables(spec, gpu_ctx, orc):
    """k=4, W=30: odds + count tables exceed one CU's LDS -> column-sliced kernels with r in HBM."""
    c = Case(**spec)
    em, ss, kmer, off, vbg = make_em(gpu_ctx, c, orc, optimizeQ=True)
    v, q = c.v0.copy(), c.q
    # the reference sums Z sequentially in fp32 (EM.cpp:179-182): over thousands of windows that sum is
    # itself ~1e-5 off (one-sided rounding), and every r of the sequence carries the factor
    long_seq = max(1.0, 4e-4 * c.L0)
    for it in range(2):
        v = em.getV(); q = em.getQ()
        em.EStep()
        s_o = orc.linear_s(v, vbg, c.K, c.W, 2)
        r_o, llh_o = orc.estep(kmer, off, c.K, c.W, s_o, q)
        np.testing.assert_allclose(em.getR(), r_o, rtol=R_RTOL * long_seq, atol=R_ATOL)
        np.testing.assert_allclose(em.getLLH(), llh_o, rtol=LLH_RTOL, atol=max(5e-7, 2e-9 * c.L0) * c.N)
        em.MStep()
        n_o = orc.mstep_counts(kmer, off, c.K, c.W, r_o)
        np.testing.assert_allclose(em.getCounts(), n_o, rtol=1e-5 * long_seq, atol=1e-6)
        np.testing.assert_allclose(em.getV(), orc.update_v(n_o, c.A, vbg, c.K, c.W), rtol=V_RTOL * long_seq, atol=1e-9)
    em.iterate(2)
    assert em.iteration() == 4
    em.close(); ss.close()


@pytest.mark.parametrize("spec", SLICED_CASES[:3], ids=[d["name"] for d in SLICED_CASES[:3]])
def test_sliced_paths_agree_bit_for_bit(spec, gpu_ctx, orc):
    """k >= 4 with the whole odds table in LDS: the E pass hands the M slices compacted lists of the windows with a
    non-zero fixed-point addend (adaptive_lists = 0: in every pass; default: chosen per pass on the device from the
    previous pass's count of non-zero windows, the first pass dense) or all responsibilities (e_list = 0); with
    e_fused = 0 the E chain itself is cut into column ranges.  The whole-table variants add the same integers: identical counts."""
    c = Case(**spec)
    res = {}
    for tag, tune in (("list", dict(adaptive_lists=0)), ("adaptive", {}), ("never_lists", dict(list_threshold_pct=0)),
                      ("lists_from_pass_2", dict(list_threshold_pct=100)), ("dense", dict(e_list=0)), ("e_sliced", dict(e_fused=0))):
        gpu_ctx.set_tuning(**tune)
        try:
            em, ss, kmer, off, vbg = make_em(gpu_ctx, c, orc, optimizeQ=True)
        finally:
            gpu_ctx.set_tuning(e_list=1, e_fused=1, adaptive_lists=1, list_threshold_pct=45)
        em.iterate(3)
        res[tag] = (em.getCounts(), em.getV(), em.getQ(), em.trace()[0], em.getR())
        em.close(); ss.close()
    for tag in ("dense", "adaptive", "never_lists", "lists_from_pass_2"):
        for k in range(5):
            assert np.array_equal(res["list"][k], res[tag][k]), (tag, k)
    np.testing.assert_allclose(res["list"][1], res["e_sliced"][1], rtol=2e-6, atol=1e-10)
    np.testing.assert_allclose(res["list"][4], res["e_sliced"][4], rtol=1e-5, atol=1e-12)


def test_optimize_stopping_rule(gpu_ctx, orc):
    c = Case(**SMALL_CASES[6])
    em, ss, kmer, off, vbg = make_em(gpu_ctx, c, orc)
    it = em.optimize()
    res = orc.optimize(kmer, off, c.K, c.W, c.bg_order, vbg, c.A, c.v0, c.q)
    # EM.cpp:117-118: the count can only flip when v_diff lands within rounding of epsilon
    assert abs(it - res["iterations"]) <= 1
    m = min(it, res["iterations"])
    llh, vd, _ = em.trace()
    np.testing.assert_allclose(llh[:m], res["trace_llh"][:m], rtol=2e-5)
    if it == res["iterations"]:
        np.testing.assert_allclose(em.getV(), res["v"], rtol=2e-4, atol=1e-8)
    em.close(); ss.close()


@pytest.mark.parametrize("stop_at", [1, 3, 12])
def test_optimize_leaves_exactly_the_state_of_its_last_pass(stop_at, gpu_ctx, orc):
    """optimize() runs one pass ahead of its stop rule; the pass enqueued behind the one that fires the rule must do
    nothing, on the device (model, q, accumulator, trace) and in the host's bookkeeping (which odds table and q slot
    are current): the handle is bit-identical to one that ran the same number of passes through iterate(), and
    stays so through further passes, an E-step and getR()."""
    c = Case(**SMALL_CASES[0])
    # a: iterate() stop_at passes; b: optimize() with a rule that fires in pass stop_at (epsilon above every v_diff
    # from that pass on: taken from a's trace); c: optimize() that runs into max_iterations = stop_at
    em_a, ss, kmer, off, vbg = make_em(gpu_ctx, c, orc, max_iterations=40)
    em_a.iterate(stop_at)
    vd = em_a.trace()[1]
    eps = float(np.nextafter(np.float32(vd[stop_at - 1]), np.float32(np.inf)))
    if stop_at > 1 and not all(v >= eps for v in vd[:stop_at - 1]):
        pytest.skip("v_diff is not decreasing on this case: no epsilon stops exactly there")
    em_b = bm.EM(gpu_ctx, ss, c.K, c.W, vbg, c.A, c.v0, c.q, bg_order=c.bg_order, max_iterations=40, epsilon=eps)
    em_c = bm.EM(gpu_ctx, ss, c.K, c.W, vbg, c.A, c.v0, c.q, bg_order=c.bg_order, max_iterations=stop_at, epsilon=0.0)
    assert em_b.optimize() == stop_at and em_c.optimize() == stop_at
    for em in (em_b, em_c):
        assert em.iteration() == stop_at
        assert np.array_equal(em.getV(), em_a.getV()) and em.getQ() == em_a.getQ()
        assert np.array_equal(em.getCounts(), em_a.getCounts())
        assert np.array_equal(em.trace()[0], em_a.trace()[0])
        assert np.array_equal(em.getR(), em_a.getR())            # the last E pass's odds table and q
    em_a.iterate(2); em_b.iterate(2)
    assert np.array_equal(em_b.getV(), em_a.getV())
    em_a.EStep(); em_b.EStep()
    assert em_b.getLLH() == em_a.getLLH() and np.array_equal(em_b.getR(), em_a.getR())
    em_a.MStep(); em_b.MStep()
    assert np.array_equal(em_b.getV(), em_a.getV())
    for em in (em_a, em_b, em_c):
        em.close()
    ss.close()


def test_optimize_q_entry_point(gpu_ctx, orc):
    c = Case(**SMALL_CASES[0])
    em, ss, kmer, off, vbg = make_em(gpu_ctx, c, orc)
    em.EStep()
    s_o = orc.linear_s(c.v0, vbg, c.K, c.W, min(c.bg_order, c.K))
    r_o, _ = orc.estep(kmer, off, c.K, c.W, s_o, c.q)
    em.optimize_q()
    np.testing.assert_allclose(em.getQ(), orc.optimize_q(r_o, off, c.W), rtol=1e-5)     # EM.cpp:515
    em.close(); ss.close()


def test_mstep_without_estep_is_a_state_error(gpu_ctx, orc):
    c = Case(**SMALL_CASES[0])
    em, ss, *_ = make_em(gpu_ctx, c, orc)
    with pytest.raises(bm.abi.BammError) as e:
        em.MStep()
    assert e.value.code == bm.abi.ERR_STATE
    em.close(); ss.close()


@pytest.mark.parametrize("spec", SMALL_CASES, ids=[d["name"] for d in SMALL_CASES])
def test_logodds_bit_exact(spec, gpu_ctx, orc):
    c = Case(**spec)
    seq, kmer, off, vbg = c.encode(orc)
    pk = bm.PackedSeqs.from_kmers(kmer, off)
    ss = bm.SeqSet(gpu_ctx, pk)
    s_log = orc.log_s(c.v0, vbg, c.K, c.W, min(c.bg_order, c.K))
    mops_o, zoops_o, z_o = orc.logodds(kmer, off, c.K, c.W, s_log)
    mops, zoops, z = bm.logodds(gpu_ctx, ss, c.K, c.W, c.bg_order, c.v0, vbg)
    assert np.array_equal(mops, mops_o)          # same fp32 add order (ScoreSeqSet.cpp:49-54)
    assert np.array_equal(zoops, zoops_o)
    assert np.array_equal(z, z_o)                # first arg-max (strict '>', ScoreSeqSet.cpp:59)
    _, zoops2, z2 = bm.logodds(gpu_ctx, ss, c.K, c.W, c.bg_order, c.v0, vbg, want_mops=False)
    assert np.array_equal(zoops2, zoops_o) and np.array_equal(z2, z_o)
    ss.close()


def test_mask_equals_subset(gpu_ctx, orc):
    """CV folds (FDR.cpp:49-57) pass a mask over a shared resident set."""
    c = Case(**SMALL_CASES[0])
    seq, kmer, off, vbg = c.encode(orc)
    pk = bm.PackedSeqs.from_kmers(kmer, off)
    ss = bm.SeqSet(gpu_ctx, pk)
    mask = (np.arange(c.N) % 4 != 1).astype(np.uint8)
    em = bm.EM(gpu_ctx, ss, c.K, c.W, vbg, c.A, c.v0, c.q, mask=mask)
    em.iterate(2)
    keep = np.nonzero(mask)[0]
    lens = np.diff(off.astype(np.int64))
    sub_kmer = np.concatenate([kmer[int(off[n]):int(off[n + 1])] for n in keep])
    sub_off = np.concatenate([[0], np.cumsum(lens[keep])]).astype(np.uint64)
    res = orc.optimize(sub_kmer, sub_off, c.K, c.W, c.bg_order, vbg, c.A, c.v0, c.q, epsilon=0.0, max_iter=2)
    np.testing.assert_allclose(em.getV(), res["v"], rtol=2e-5, atol=1e-9)
    em.close(); ss.close()


def test_two_shards_with_allreduce_callback_equal_one(gpu_ctx, orc):
    """Sequences split over two handles + a summing callback == one handle."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    c = Case(**SMALL_CASES[6])
    seq, kmer, off, vbg = c.encode(orc)
    pk = bm.PackedSeqs.from_kmers(kmer, off)
    b0, e0 = pk.shard_range(c.W, 0, 2)
    b1, e1 = pk.shard_range(c.W, 1, 2)
    assert (b0, e1) == (0, c.N) and e0 == b1
    sa, sb, sall = bm.SeqSet(gpu_ctx, pk, b0, e0), bm.SeqSet(gpu_ctx, pk, b1, e1), bm.SeqSet(gpu_ctx, pk)
    ea = bm.EM(gpu_ctx, sa, c.K, c.W, vbg, c.A, c.v0, c.q, optimizeQ=True)
    eb = bm.EM(gpu_ctx, sb, c.K, c.W, vbg, c.A, c.v0, c.q, optimizeQ=True)
    eall = bm.EM(gpu_ctx, sall, c.K, c.W, vbg, c.A, c.v0, c.q, optimizeQ=True)
    pa, n = ea.reduce_buffer()
    pb, _ = eb.reduce_buffer()
    ha, hb = np.zeros(n, np.int64), np.zeros(n, np.int64)          # the accumulator holds 64-bit integers
    for _ in range(3):
        ea.accumulate(); eb.accumulate()
        gpu_ctx.sync()
        assert hip.hipMemcpy(ha.ctypes.data_as(C.c_void_p), C.c_void_p(pa), n * 8, 2) == 0
        assert hip.hipMemcpy(hb.ctypes.data_as(C.c_void_p), C.c_void_p(pb), n * 8, 2) == 0
        tot = ha + hb
        assert hip.hipMemcpy(C.c_void_p(pa), tot.ctypes.data_as(C.c_void_p), n * 8, 1) == 0
        assert hip.hipMemcpy(C.c_void_p(pb), tot.ctypes.data_as(C.c_void_p), n * 8, 1) == 0
        ea.update(); eb.update()
    eall.iterate(3)
    assert np.array_equal(ea.getV(), eb.getV())                  # redundant, identical updates
    assert np.array_equal(ea.getV(), eall.getV())                # integer sums: the split does not show
    assert np.array_equal(ea.getCounts(), eall.getCounts())
    assert ea.getQ() == eall.getQ() and np.array_equal(ea.trace()[0], eall.trace()[0])   # per-sequence rounding: exact sums
    for x in (ea, eb, eall, sa, sb, sall):
        x.close()


def test_empty_set_and_errors(gpu_ctx, orc):
    pk = bm.PackedSeqs.from_kmers(np.zeros(0, np.uint64), np.zeros(1, np.uint64))
    ss = bm.SeqSet(gpu_ctx, pk)
    c = Case(**SMALL_CASES[0])
    _, _, _, vbg = c.encode(orc)
    em = bm.EM(gpu_ctx, ss, c.K, c.W, vbg, c.A, c.v0, c.q)
    em.iterate(1)
    n = em.getCounts()
    assert np.all(n == 0)
    # with zero counts updateV returns the prior chain (Motif.h:110-135)
    v_o = orc.update_v(n, c.A, vbg, c.K, c.W)
    np.testing.assert_allclose(em.getV(), v_o, rtol=1e-6)
    em.close(); ss.close()
    # a sequence shorter than the motif is refused (mainBaMM.cpp:75-83 drops them beforehand)
    c2 = Case("short", N=4, L0=5, W=8, K=1, ss=True)
    _, kmer, off, vbg2 = c2.encode(orc)
    ss2 = bm.SeqSet(gpu_ctx, bm.PackedSeqs.from_kmers(kmer, off))
    with pytest.raises(bm.abi.BammError) as e:
        bm.EM(gpu_ctx, ss2, c2.K, c2.W, vbg2, c2.A, c2.v0, c2.q)
    assert e.value.code == bm.abi.ERR_ARG
    ss2.close()


EXTRA_SHAPES = [
    dict(name="x_long_M32", N=6, L0=1900, W=25, K=2, ss=True, ragged=140, n_frac=0.002),      # M classes 28/32
    dict(name="x_long_ds_M24", N=8, L0=720, W=17, K=1, ragged=60),                             # ds -> L ~ 1441: M = 24
    dict(name="x_w_odd_k3", N=40, L0=90, W=13, K=3, ragged=20, n_frac=0.01),                   # W not a multiple of 4
    dict(name="x_w3_k0", N=50, L0=30, W=3, K=0, ss=True, ragged=5),
    dict(name="x_w_gt_L", N=20, L0=24, W=24, K=2, ss=True),                                    # exactly one window
    dict(name="x_bg_order0", N=60, L0=64, W=9, K=2, bg_order=0),
    dict(name="x_bg_order3_k1", N=60, L0=64, W=9, K=1, bg_order=3),                            # K_bg = min(3, 1)
    dict(name="x_all_lengths", N=130, L0=330, W=10, K=2, ss=True, ragged=300),                 # many M buckets in one set
    dict(name="x_M12_14_16", N=40, L0=820, W=21, K=2, ss=True, ragged=110, n_frac=0.003),      # 512-thread classes
    dict(name="x_M20_28_ds", N=24, L0=760, W=20, K=2, ragged=130),                             # ds: L 1261..1781
    dict(name="x_M40_64", N=12, L0=3300, W=18, K=2, ss=True, ragged=790, n_frac=0.001),        # L 2510..4090
    dict(name="x_M80_128", N=9, L0=6150, W=16, K=2, ss=True, ragged=2040, n_frac=0.0005),      # L 4110..8190: one wave per SIMD
    dict(name="x_M128_ds", N=3, L0=4040, W=22, K=1, ragged=50, n_frac=0.0005),                 # ds: L 7981..8181
]


@pytest.mark.parametrize("spec", EXTRA_SHAPES, ids=[d["name"] for d in EXTRA_SHAPES])
def test_shapes_and_length_buckets(spec, gpu_ctx, orc):
    """Every kernel instantiation (M = 1..32 positions per lane), odd widths, bg orders above and
    below the motif order, single-window sequences, mixed-length sets."""
    c = Case(**spec)
    em, ss, kmer, off, vbg = make_em(gpu_ctx, c, orc, optimizeQ=True)
    Kb = min(c.bg_order, c.K)
    em.EStep()
    s_o = orc.linear_s(c.v0, vbg, c.K, c.W, Kb)
    r_o, llh_o = orc.estep(kmer, off, c.K, c.W, s_o, c.q)
    np.testing.assert_allclose(em.getR(), r_o, rtol=2e-5, atol=R_ATOL)
    # the oracle sums Z sequentially in fp32: ~sqrt(L)*6e-8 relative noise per sequence
    np.testing.assert_allclose(em.getLLH(), llh_o, rtol=LLH_RTOL, atol=1e-8 * float(off[-1]))
    em.MStep()
    n_o = orc.mstep_counts(kmer, off, c.K, c.W, r_o)
    np.testing.assert_allclose(em.getCounts(), n_o, rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(em.getV(), orc.update_v(n_o, c.A, vbg, c.K, c.W), rtol=2e-5, atol=1e-9)
    em.iterate(3)
    res = orc.optimize(kmer, off, c.K, c.W, c.bg_order, vbg, c.A, em.getV() * 0 + c.v0, c.q, optimizeQ=True,
                       epsilon=0.0, max_iter=4)
    # 1 (EStep/MStep without q update) + 3 fused passes vs 4 oracle passes differ only in the
    # first pass's q update; compare the scorer instead on the device's own model
    v = em.getV()
    s_log = orc.log_s(v, vbg, c.K, c.W, Kb)
    mops_o, zoops_o, z_o = orc.logodds(kmer, off, c.K, c.W, s_log)
    mops, zoops, z = bm.logodds(gpu_ctx, ss, c.K, c.W, c.bg_order, v, vbg)
    assert np.array_equal(mops, mops_o) and np.array_equal(zoops, zoops_o) and np.array_equal(z, z_o)
    em.close(); ss.close()


HIGH_ORDER_CASES = [
    dict(name="k7_w5", N=40, L0=70, W=5, K=7, ragged=15, n_frac=0.01),
    dict(name="k8_w3_ss", N=30, L0=90, W=3, K=8, ss=True, ragged=10),
    dict(name="k10_w2", N=12, L0=60, W=2, K=10, ragged=8, n_frac=0.02),
]


@pytest.mark.parametrize("spec", HIGH_ORDER_CASES, ids=[d["name"] for d in HIGH_ORDER_CASES])
def test_orders_beyond_the_lds_envelope(spec, gpu_ctx, orc):
    """Orders 7..10 (kmer_ spans 11 bases, Sequence.cpp:37; BaMM files go to order 8, MotifSet.cpp:202-205): not one
    column of the 4^(K+1)-row tables fits a CU's LDS, so the pass runs with its tables in global memory
    (csrc/long_seq.hip).  E-step, M-step, updateV, getR and the scorer against the oracle, as for every other order."""
    c = Case(**spec)
    em, ss, kmer, off, vbg = make_em(gpu_ctx, c, orc, optimizeQ=True)
    Kb = min(c.bg_order, c.K)
    for it in range(2):
        v, q = em.getV(), em.getQ()
        em.EStep()
        s_o = orc.linear_s(v, vbg, c.K, c.W, Kb)
        assert np.array_equal(em.getS(), s_o)
        r_o, llh_o = orc.estep(kmer, off, c.K, c.W, s_o, q)
        np.testing.assert_allclose(em.getR(), r_o, rtol=R_RTOL, atol=R_ATOL)
        np.testing.assert_allclose(em.getLLH(), llh_o, rtol=LLH_RTOL, atol=5e-7 * c.N)
        em.MStep()
        n_o = orc.mstep_counts(kmer, off, c.K, c.W, r_o)
        np.testing.assert_allclose(em.getCounts(), n_o, rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(em.getV(), orc.update_v(n_o, c.A, vbg, c.K, c.W), rtol=V_RTOL, atol=1e-9)
    em.close()
    em, ss2, *_ = make_em(gpu_ctx, c, orc, optimizeQ=True)     # EM::optimize's loop (q re-estimated in its first passes)
    em.iterate(4)
    assert em.iteration() == 4
    res = orc.optimize(kmer, off, c.K, c.W, c.bg_order, vbg, c.A, c.v0, c.q, optimizeQ=True, epsilon=0.0, max_iter=4)
    np.testing.assert_allclose(em.getV(), res["v"], rtol=5e-5, atol=1e-8)
    np.testing.assert_allclose(em.getQ(), res["q"], rtol=1e-5)
    ss2.close()
    v = em.getV()
    mops_o, zoops_o, z_o = orc.logodds(kmer, off, c.K, c.W, orc.log_s(v, vbg, c.K, c.W, Kb))
    mops, zoops, z = bm.logodds(gpu_ctx, ss, c.K, c.W, c.bg_order, v, vbg)
    assert np.array_equal(mops, mops_o) and np.array_equal(zoops, zoops_o) and np.array_equal(z, z_o)
    em.close(); ss.close()


LONG_SPECS = [EXTRA_SHAPES[-2], EXTRA_SHAPES[-1], SLICED_CASES[1]]


@pytest.mark.parametrize("spec", LONG_SPECS, ids=[d["name"] for d in LONG_SPECS])
def test_long_sequences_match_exact_arithmetic(spec, gpu_ctx, orc):
    """L = 4097..8192 (80, 96 and 128 positions per lane, one wave per SIMD): against the fp32 reference
    arithmetic these lengths only allow a loose bar (its sequential Z sum), against the fp64 restatement of
    the same formulas the device is within 1e-6."""
    c = Case(**spec)
    em, ss, kmer, off, vbg = make_em(gpu_ctx, c, orc)
    v64, n64, llh64, _ = orc.em_step_f64(kmer, off, c.K, c.W, c.bg_order, vbg, c.A, c.v0, c.q)
    em.iterate(1)
    np.testing.assert_allclose(em.getV(), v64, rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(em.getCounts(), n64, rtol=2e-6, atol=1e-7)
    llh, _, _ = em.trace()
    np.testing.assert_allclose(llh[-1], llh64, rtol=1e-6)
    em.close(); ss.close()


def test_sequences_beyond_the_length_classes(gpu_ctx, orc):
    """More than 8192 positions: accepted (csrc/long_seq.hip, tests/test_long_gpu.py checks the numbers).  EM::mask and
    PWM seeding have no limit either: their per-wave arrays move to a global scratch region (mask: with 32-bit window
    lists there; tests/test_mask_gpu.py::m_wide_lists and tests/test_seed_gpu.py check the numbers)."""
    c = Case("toolong", N=2, L0=70000, W=8, K=1, ss=True)
    _, kmer, off, vbg = c.encode(orc)
    ss = bm.SeqSet(gpu_ctx, bm.PackedSeqs.from_kmers(kmer, off))
    assert ss.info()["max_len"] == 70000
    em = bm.EM(gpu_ctx, ss, c.K, c.W, vbg, c.A, c.v0, c.q)
    assert em.plan()[2] == 1                                  # one launch: the long bucket
    res = orc.mask(kmer, off, c.K, c.W, c.bg_order, vbg, c.A, c.v0, c.q, f=0.1, epsilon=0.0, max_iter=1)
    em_m = bm.EM(gpu_ctx, ss, c.K, c.W, vbg, c.A, c.v0, c.q, epsilon=0.0, max_iterations=1)
    em_m.mask(0.1)                                            # 70 000 positions: beyond 16-bit window indices
    assert np.float32(em_m.last_mask["cutoff"]) == np.float32(res["cutoff"]) and em_m.last_mask["listed"] == res["listed"]
    em_m.close()
    counts, z = bm.seed_from_pwm(gpu_ctx, ss, c.K, c.W, np.ones(4 * c.W, np.float32), 0.3, np.full(c.N, 0.85))
    LW1 = 70000 - c.W + 1                                     # flat odds: 0.7 on "no motif", 0.3 / LW1 per window
    assert np.all(np.abs(z.astype(np.int64) - LW1 // 2) < LW1 // 100)
    assert np.all(counts[: 4 * c.W].reshape(4, c.W).sum(axis=0) == c.N)
    em.close(); ss.close()


def test_logodds_subset(gpu_ctx, orc):
    c = Case(**SMALL_CASES[0])
    seq, kmer, off, vbg = c.encode(orc)
    ss = bm.SeqSet(gpu_ctx, bm.PackedSeqs.from_kmers(kmer, off))
    mask = (np.arange(c.N) % 4 == 2).astype(np.uint8)
    full = bm.logodds(gpu_ctx, ss, c.K, c.W, c.bg_order, c.v0, vbg)
    sub = bm.logodds(gpu_ctx, ss, c.K, c.W, c.bg_order, c.v0, vbg, mask=mask)
    sel = mask.astype(bool)
    assert np.array_equal(sub[1][sel], full[1][sel]) and np.array_equal(sub[2][sel], full[2][sel])
    assert np.all(sub[1][~sel] == 0) and np.all(sub[2][~sel] == 0)
    lens = np.diff(off.astype(np.int64)) - c.W + 1
    moff = np.concatenate([[0], np.cumsum(lens)])
    for n in range(c.N):
        a, b = sub[0][moff[n]:moff[n + 1]], full[0][moff[n]:moff[n + 1]]
        assert np.array_equal(a, b) if sel[n] else np.all(a == 0)
    ss.close()


def test_external_reduce_buffer(gpu_ctx, orc):
    """The fused buffer can live in caller-owned device memory (what bench.py does with a torch
    tensor for RCCL)."""
    import torch
    c = Case(**SMALL_CASES[0])
    em, ss, kmer, off, vbg = make_em(gpu_ctx, c, orc)
    _, n = em.reduce_buffer()
    red = torch.zeros(n + 5, dtype=torch.int64, device="cuda:0")
    torch.cuda.synchronize()
    em.set_reduce_buffer(red.data_ptr(), n + 5)
    em.iterate(1)
    em.accumulate()                                          # second pass, stopped before its update
    gpu_ctx.sync()
    res = orc.optimize(kmer, off, c.K, c.W, c.bg_order, vbg, c.A, c.v0, c.q, epsilon=0.0, max_iter=2)
    host = red.cpu().numpy()
    cells = 4 ** (c.K + 1) * c.W
    np.testing.assert_allclose(host[cells] * 2.0 ** -24, res["llh"], rtol=1e-5)          # llh of the second pass
    np.testing.assert_allclose(host[:cells].reshape(-1, c.W) * 2.0 ** -40,
                               res["n"][bm.v_offset(c.K, c.W):].reshape(-1, c.W), rtol=2e-5, atol=1e-6)
    assert host[cells + 2] == c.N and np.all(host[n:] == 0)
    em.update()
    gpu_ctx.sync()
    assert np.all(red.cpu().numpy() == 0)                     # consumed and cleared for the next pass
    np.testing.assert_allclose(em.getV(), res["v"], rtol=2e-5, atol=1e-9)
    with pytest.raises(bm.abi.BammError):
        em.set_reduce_buffer(red.data_ptr(), 3)
    em.close(); ss.close()


def test_kernel_timing_modes(gpu_ctx, orc):
    """bamm_em_set_kernel_timing: a pair of HIP events around every n-th pass, none, or ONE pair around all passes of a call
    (BAMM_TIMING_WHOLE_CALL) -- `launches` counts the passes an interval covers, the whole-call interval lies inside the host's
    clock around the call, and the timing mode changes nothing in the model."""
    import time
    c = Case(**SMALL_CASES[6])
    models = []
    for every, want in ((1, 12), (4, 3), (0, 0), (-1, 12)):
        em, ss, kmer, off, vbg = make_em(gpu_ctx, c, orc, max_iterations=40)
        em.set_kernel_timing(every)
        em.iterate(3)                                         # an earlier call's events are not carried over
        gpu_ctx.sync()
        t0 = time.perf_counter()
        em.iterate(12)
        gpu_ctx.sync()
        host_ms = (time.perf_counter() - t0) * 1e3
        ms, n = em.kernel_time()
        assert n == want, (every, n)
        if want:
            assert 0.0 < ms <= host_ms * 1.05, (every, ms, host_ms)
        models.append(em.getV().copy())
        if every == -1:
            em.optimize()                                     # look-ahead passes that found the stop flag raised are launches too
            ms2, n2 = em.kernel_time()
            assert n2 >= em.iteration() - 15 and ms2 > 0.0
        em.close(); ss.close()
    assert all(np.array_equal(models[0], m) for m in models[1:])
