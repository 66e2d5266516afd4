"""EM::mask (`--advanceEM`, EM.cpp:261-503) on the device against the oracle's restatement, which is
bit-identical to the reference's own mask() (tests/test_mask_cpu.py).

The order-0 pass sums in the reference's sequential fp32 order, so the cut-off, the list of
windows and (with optimizeQ) the per-sequence q chain must match EXACTLY; the masked iterations
carry summation-order noise only.
"""
import numpy as np
import pytest

import bammmotif2_amd as bm
from tests import margins
from tests.cases import SMALL_CASES, Case
from tests.test_parity_gpu import make_em

pytestmark = pytest.mark.gpu

CASES = [d for d in SMALL_CASES if d["W"] >= 2] + [
    dict(name="m_long", N=6, L0=6100, W=12, K=2, ss=True, ragged=2000, n_frac=0.0005),      # M = 80..128
    # beyond the per-wave LDS arrays (~16 000 positions at k = 2): the arrays live in a global scratch region per wave
    dict(name="m_xlong", N=5, L0=14000, W=10, K=2, ragged=9000, n_frac=0.0002),
    dict(name="m_xlong_k4", N=4, L0=30000, W=8, K=4, ss=True, ragged=12000),
    # beyond 65 535 positions (both strands of 40 000 bases): the window lists in the scratch region are 32 bits wide
    dict(name="m_wide_lists", N=3, L0=40000, W=10, K=1, ragged=3000, n_frac=0.0002),
    # order 7: one column of the count table (4^8 cells of 8 bytes) exceeds the LDS -- the listed windows' addends go
    # straight into the accumulator
    dict(name="m_k7", N=60, L0=260, W=6, K=7, ragged=60, n_frac=0.002),
    dict(name="m_k8_ss", N=30, L0=400, W=5, K=8, ss=True)]


@pytest.mark.parametrize("oq", [False, True], ids=["fixq", "optq"])
@pytest.mark.parametrize("f", [0.05, 0.2])
@pytest.mark.parametrize("spec", CASES, ids=[d["name"] for d in CASES])
def test_mask_three_passes_match_oracle(spec, f, oq, gpu_ctx, orc):
    c = Case(**spec)
    em, ss, kmer, off, vbg = make_em(gpu_ctx, c, orc, optimizeQ=oq, epsilon=0.0, max_iterations=3)
    it = em.mask(f)
    res = orc.mask(kmer, off, c.K, c.W, c.bg_order, vbg, c.A, c.v0, c.q, optimizeQ=oq, f=f, epsilon=0.0, max_iter=3)
    assert it == res["iterations"] == 3
    assert np.float32(em.last_mask["cutoff"]) == np.float32(res["cutoff"])          # EM.cpp:343
    assert em.last_mask["listed"] == res["listed"]                                   # EM.cpp:345-356
    assert np.float32(em.getQ()) == np.float32(res["q"])                             # EM.cpp:321 chain
    name, fl = f"mask {c.name} f={f} optQ={int(oq)}", "mask kernels"
    # flat 1e-5; the restatement's normaliser is a sequential fp32 sum over a sequence's listed windows (EM.cpp:417), whose
    # own rounding grows with their number: beyond 4 000 listed windows per sequence (m_wide_lists at f = 0.2: 16 000)
    # the bar follows sqrt(count)
    flat = 1e-5 * max(1.0, float(np.sqrt(res["listed"] / c.N / 4000.0)))
    margins.check(name, fl, "n pass 3", em.getCounts(), res["n"], flat, 1e-7, against="fp32 restatement")
    margins.check(name, fl, "v pass 3", em.getV(), res["v"], flat, 1e-9, against="fp32 restatement")
    llh, vd, _ = em.trace()
    # the reference's llh is a sequential fp32 sum over the sequences: ~1e-7 per term
    np.testing.assert_allclose(llh, res["trace_llh"], rtol=1e-5, atol=max(1e-5, 3e-7 * c.N))
    # v_diff is the reference's sequential fp32 sum of |dv| over the whole model (EM.cpp:102-108): 1.7 M terms at order 8
    np.testing.assert_allclose(vd, res["trace_vdiff"], rtol=1e-3 * max(1.0, float(np.sqrt(c.v0.size / 1e5))), atol=1e-6)
    r = em.getR()
    # incl. the r_[n][0] decay (:421).  Flat 1e-5 on n, v and r: the observed margins (profiles/r04_parity_margins.txt)
    # stay below 5e-6 on every case
    margins.check(name, fl, "r pass 3", r, res["r"], flat, 1e-12, against="fp32 restatement")
    assert np.array_equal(r == 0, res["r"] == 0)
    em.close(); ss.close()


def test_mask_stopping_rule_and_state_checks(gpu_ctx, orc):
    c = Case(**SMALL_CASES[0])
    em, ss, kmer, off, vbg = make_em(gpu_ctx, c, orc)
    it = em.mask(0.05)
    res = orc.mask(kmer, off, c.K, c.W, c.bg_order, vbg, c.A, c.v0, c.q)
    m = min(it, res["iterations"])
    assert m >= 5
    llh, vd, _ = em.trace()
    np.testing.assert_allclose(llh[:m], res["trace_llh"][:m], rtol=2e-5, atol=1e-5)
    if it == res["iterations"]:
        np.testing.assert_allclose(em.getV(), res["v"], rtol=2e-4, atol=1e-8)
    with pytest.raises(RuntimeError, match="already run"):
        em.mask(0.05)
    em.close()
    em2 = bm.EM(gpu_ctx, ss, c.K, c.W, vbg, c.A, c.v0, c.q, bg_order=c.bg_order)
    with pytest.raises(RuntimeError, match="fraction"):
        em2.mask(1.5)
    em2.close(); ss.close()


def test_mask_on_a_fold(gpu_ctx, orc):
    """FDR.cpp:49-70: mask() on the training part of a fold == mask() on that subset alone."""
    c = Case(**SMALL_CASES[0])
    seq, kmer, off, vbg = c.encode(orc)
    keep = (np.arange(c.N) % 4 != 1)
    pk = bm.PackedSeqs.from_kmers(kmer, off)
    ss = bm.SeqSet(gpu_ctx, pk)
    em = bm.EM(gpu_ctx, ss, c.K, c.W, vbg, c.A, c.v0, c.q, bg_order=c.bg_order, optimizeQ=True,
               mask=keep.astype(np.uint8), epsilon=0.0, max_iterations=2)
    em.mask(0.1)
    lens = np.diff(off.astype(np.int64))
    sub_kmer = np.concatenate([kmer[int(off[n]):int(off[n + 1])] for n in range(c.N) if keep[n]])
    sub_off = np.concatenate([[0], np.cumsum(lens[keep])]).astype(np.uint64)
    res = orc.mask(sub_kmer, sub_off, c.K, c.W, c.bg_order, vbg, c.A, c.v0, c.q, optimizeQ=True, f=0.1,
                   epsilon=0.0, max_iter=2)
    assert np.float32(em.last_mask["cutoff"]) == np.float32(res["cutoff"])
    assert em.last_mask["listed"] == res["listed"]
    assert np.float32(em.getQ()) == np.float32(res["q"])
    np.testing.assert_allclose(em.getV(), res["v"], rtol=2e-5, atol=1e-9)
    em.close(); ss.close()


def test_mask_k4_sliced_counts(gpu_ctx, orc):
    """k=4, W=30: odds table read from HBM, counts in column slices."""
    c = Case("k4", N=60, L0=120, W=30, K=4, seed=5, ss=True)
    em, ss, kmer, off, vbg = make_em(gpu_ctx, c, orc, epsilon=0.0, max_iterations=2)
    em.mask(0.1)
    res = orc.mask(kmer, off, c.K, c.W, c.bg_order, vbg, c.A, c.v0, c.q, f=0.1, epsilon=0.0, max_iter=2)
    assert em.last_mask["listed"] == res["listed"]
    name, fl = "mask k4 sliced f=0.1", "mask kernels"
    margins.check(name, fl, "n pass 2", em.getCounts(), res["n"], 1e-5, 1e-7, against="fp32 restatement")
    margins.check(name, fl, "v pass 2", em.getV(), res["v"], 1e-5, 1e-9, against="fp32 restatement")
    em.close(); ss.close()


def _mask_fixtures():
    import glob, os
    from tests.golden_util import GOLDEN_DIR
    return sorted(os.path.basename(p)[5:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "mask_*.npz")))


@pytest.mark.parametrize("name", _mask_fixtures())
def test_mask_against_reference_golden(name, gpu_ctx, orc):
    """The HIP path against values the reference's own mask() produced (tests/golden/mask_*.npz)."""
    import os
    from tests.golden_util import GOLDEN_DIR
    c = Case(**next(d for d in SMALL_CASES if d["name"] == name))
    g = np.load(os.path.join(GOLDEN_DIR, f"mask_{name}.npz"))
    for oq in (0, 1):
        for f in (0.05, 0.2):
            t = f"oq{oq}_f{int(f * 100)}"
            em, ss, kmer, off, vbg = make_em(gpu_ctx, c, orc, optimizeQ=bool(oq))
            it = em.mask(f)
            assert np.float32(em.last_mask["cutoff"]) == g[t + "_cutoff"]
            assert em.last_mask["listed"] == int(g[t + "_listed"])
            assert np.float32(em.getQ()) == g[t + "_q"]
            m = min(it, int(g[t + "_iterations"]))
            llh, vd, _ = em.trace()
            # the trajectory is compared while it is still well-conditioned; EM amplifies rounding
            # noise over dozens of passes (see tests/test_golden_gpu.py)
            head = min(m, 8)
            np.testing.assert_allclose(llh[:head], g[t + "_trace_llh"][:head], rtol=2e-5, atol=1e-5)
            np.testing.assert_allclose(vd[:head], g[t + "_trace_vdiff"][:head], rtol=2e-3, atol=1e-5)
            if it == int(g[t + "_iterations"]):
                np.testing.assert_allclose(em.getV(), g[t + "_v"], rtol=1e-3, atol=1e-6)
                nr = len(g[t + "_r"])
                r = em.getR()[:nr]
                np.testing.assert_allclose(r, g[t + "_r"], rtol=2e-3, atol=1e-9)
            em.close(); ss.close()


@pytest.mark.parametrize("spec", [CASES[0], [d for d in CASES if d["name"] == "m_k7"][0], [d for d in CASES if d["name"] == "m_wide_lists"][0]],
                         ids=["small", "k7_direct", "wide_lists"])
def test_mask_sharded_over_two_contexts_is_the_one_rank_mask(spec, gpu_ctx, orc):
    """EM::mask over two ranks (contexts on the one device, the host-staged communicator): the cut-off comes from the summed
    integer histogram and the counts are integer sums, so cut-off, list size and the model equal the one-rank run's bit for
    bit -- also where the ranks' plans differ (one shard holds the 83 095-position record) and where the counts go straight
    into the accumulator (order 7)."""
    import threading
    c = Case(**spec)
    one, ss1, kmer, off, vbg = make_em(gpu_ctx, c, orc, epsilon=0.0, max_iterations=3)
    it1 = one.mask(0.1)
    want = (it1, dict(one.last_mask), one.getV().copy(), one.getCounts().copy())
    one.close(); ss1.close()
    pk = bm.PackedSeqs.from_kmers(kmer, off)
    ctxs = [bm.Context(0) for _ in range(2)]
    comms = bm.Comm.init_local(ctxs, max(4 ** (c.K + 1) * c.W + 3, 2049))
    sets, ems = [], []
    for r in range(2):
        b, e = pk.shard_range(c.W, r, 2)
        ss = bm.SeqSet(ctxs[r], pk, b, e)
        em = bm.EM(ctxs[r], ss, c.K, c.W, vbg, c.A, c.v0, c.q, bg_order=c.bg_order, n_seqs_global=c.N, n_seqs_bound=c.N,
                   epsilon=0.0, max_iterations=3)
        em.set_comm(comms[r])
        sets.append(ss); ems.append(em)
    got, errs = [None, None], [None, None]

    def worker(r):
        try:
            it = ems[r].mask(0.1)
            got[r] = (it, dict(ems[r].last_mask), ems[r].getV().copy(), ems[r].getCounts().copy())
        except Exception as e:
            errs[r] = e
            for x in comms:
                x.abort()

    th = [threading.Thread(target=worker, args=(r,)) for r in range(2)]
    for t in th: t.start()
    for t in th: t.join(timeout=120)
    assert not any(t.is_alive() for t in th) and errs == [None, None], errs
    assert sum(g[1]["listed"] for g in got) == want[1]["listed"]          # a rank reports the windows IT lists
    for r in range(2):
        assert got[r][0] == want[0] and np.float32(got[r][1]["cutoff"]) == np.float32(want[1]["cutoff"])
        assert np.array_equal(got[r][2], want[2]) and np.array_equal(got[r][3], want[3])
    for x in ems + sets + comms + ctxs:
        x.close()
