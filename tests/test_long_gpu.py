"""Sequences beyond 8192 positions (csrc/long_seq.hip): the reference has no length limit
(src/init/Sequence.cpp:4-43), the register-resident kernels hold at most 128 positions per lane.  Longer
records are walked window by window; mixed with ordinary ones in one set they must give what the oracle gives."""
import numpy as np
import pytest

import bammmotif2_amd as bm
from tests.cases import Case
from tests.test_parity_gpu import LLH_RTOL, R_ATOL, V_RTOL, make_em

pytestmark = pytest.mark.gpu

LONG_CASES = [
    dict(name="long_k2_ss_mixed", N=14, L0=12000, W=12, K=2, ss=True, ragged=11900, n_frac=0.001),   # 100 .. 12000 positions
    dict(name="long_k1_ds", N=5, L0=5200, W=9, K=1, ragged=1500, n_frac=0.0005),                     # both strands: L up to 10401
    dict(name="long_k4_sliced", N=6, L0=9000, W=30, K=4, ss=True, ragged=4000, n_frac=0.0005),       # column-sliced path + long bucket
    dict(name="long_k0_only_long", N=3, L0=20000, W=6, K=0, ss=True, ragged=2000),                   # every sequence long
]


@pytest.mark.parametrize("spec", LONG_CASES, ids=[d["name"] for d in LONG_CASES])
def test_long_sequences_match_oracle(spec, gpu_ctx, orc):
    c = Case(**spec)
    em, ss, kmer, off, vbg = make_em(gpu_ctx, c, orc, optimizeQ=True)
    lens = np.diff(off.astype(np.int64))
    assert lens.max() > 8192
    Kb = min(c.bg_order, c.K)
    # the oracle sums Z sequentially in fp32 (EM.cpp:179-182): over >10^4 windows that sum is itself ~1e-5 off
    long_tol = max(1.0, 4e-4 * lens.max() / (1 if c.ss else 2))
    for it in range(2):
        v, q = em.getV(), em.getQ()
        em.EStep()
        s_o = orc.linear_s(v, vbg, c.K, c.W, Kb)
        r_o, llh_o = orc.estep(kmer, off, c.K, c.W, s_o, q)
        r_g = em.getR()
        np.testing.assert_allclose(r_g, r_o, rtol=1e-5 * long_tol, atol=R_ATOL)
        assert np.array_equal(r_g == 0, r_o == 0)
        np.testing.assert_allclose(em.getLLH(), llh_o, rtol=LLH_RTOL, atol=2e-9 * float(off[-1]))
        em.MStep()
        n_o = orc.mstep_counts(kmer, off, c.K, c.W, r_o)
        np.testing.assert_allclose(em.getCounts(), n_o, rtol=1e-5 * long_tol, atol=1e-6)
        np.testing.assert_allclose(em.getV(), orc.update_v(n_o, c.A, vbg, c.K, c.W), rtol=V_RTOL * long_tol, atol=1e-9)
    em.iterate(2)
    assert em.iteration() == 4
    # the exact-arithmetic restatement: no summation-order noise on either side
    em2, ss2, *_ = make_em(gpu_ctx, c, orc)
    v64, *_ = orc.em_step_f64(kmer, off, c.K, c.W, c.bg_order, vbg, c.A, c.v0, c.q)
    em2.iterate(1)
    np.testing.assert_allclose(em2.getV(), v64, rtol=1e-6, atol=1e-9)
    # scorer: bit-exact, long and short sequences alike
    v = em.getV()
    s_log = orc.log_s(v, vbg, c.K, c.W, Kb)
    mops_o, zo_o, z_o = orc.logodds(kmer, off, c.K, c.W, s_log)
    mops, zo, z = bm.logodds(gpu_ctx, ss, c.K, c.W, c.bg_order, v, vbg)
    assert np.array_equal(zo, zo_o) and np.array_equal(z, z_o) and np.array_equal(mops, mops_o)
    for x in (em, em2, ss, ss2):
        x.close()


def test_long_and_short_paths_add_the_same_integers(gpu_ctx, orc):
    """A set with one long record, trained as a whole and as two shards (the long record alone in one): the
    accumulator is an integer sum, so the split does not show -- the long path adds exactly what it should."""
    c = Case(name="long_split", N=40, L0=300, W=10, K=2, ss=True, ragged=100)
    long_c = Case(name="long_one", N=1, L0=9500, W=10, K=2, ss=True, seed=5)
    codes = np.concatenate([c.codes, long_c.codes])
    in_off = np.concatenate([c.in_off, c.in_off[-1] + long_c.in_off[1:]])
    _, kmer, off = orc.encode_set(codes, in_off, True, 42)
    vbg = orc.bg_model(kmer, off, 2, c.alpha_bg)
    pk = bm.PackedSeqs.from_kmers(kmer, off)
    whole = bm.SeqSet(gpu_ctx, pk)
    a, b = bm.SeqSet(gpu_ctx, pk, 0, c.N), bm.SeqSet(gpu_ctx, pk, c.N, c.N + 1)
    ew = bm.EM(gpu_ctx, whole, c.K, c.W, vbg, c.A, c.v0, c.q)
    ea = bm.EM(gpu_ctx, a, c.K, c.W, vbg, c.A, c.v0, c.q)
    eb = bm.EM(gpu_ctx, b, c.K, c.W, vbg, c.A, c.v0, c.q)
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")

    def raw(em):
        em.accumulate()
        gpu_ctx.sync()
        p, n = em.reduce_buffer()
        h = np.zeros(n, np.int64)
        assert hip.hipMemcpy(h.ctypes.data_as(C.c_void_p), C.c_void_p(p), n * 8, 2) == 0
        return h

    hw, ha, hb = raw(ew), raw(ea), raw(eb)
    cells = 4 ** (c.K + 1) * c.W
    assert np.array_equal(hw[:cells], (ha + hb)[:cells]) and hw[cells + 2] == ha[cells + 2] + hb[cells + 2] == c.N + 1
    assert hb[:cells].sum() > 0
    for x in (ew, ea, eb, whole, a, b):
        x.close()
