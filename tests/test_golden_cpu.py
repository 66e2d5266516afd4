"""The CPU oracle and the product's host helpers against the golden vectors the real reference
produced (tests/golden).  Integer / fp32-arithmetic results must match bit for bit; values that
pass through libm's logf are allowed 1 ulp-level slack (glibc picks CPU-specific variants)."""
import numpy as np
import pytest

import bammmotif2_amd as bm
from tests import golden_util as gu

NAMES = gu.fixture_names()


def test_fixtures_present():
    assert len(NAMES) >= 8, NAMES


@pytest.mark.parametrize("name", NAMES)
def test_oracle_matches_reference_golden(name, orc):
    c, g = gu.load(name)
    seq, kmer, off = orc.encode_set(c.codes, c.in_off, c.ss, 42)
    assert np.array_equal(off, g["off"])
    assert gu.digest(kmer) == str(g["kmer_sha256"])          # Sequence.cpp:35-41 incl. rand() protocol
    if "kmer" in g:
        assert np.array_equal(kmer, g["kmer"]) and np.array_equal(seq, g["seq"])
    vbg = orc.bg_model(kmer, off, c.bg_order, c.alpha_bg)
    assert np.array_equal(vbg, g["vbg"])                      # BackgroundModel.cpp:26-42,441-473
    Kb = min(c.bg_order, c.K)
    v, q = c.v0.copy(), c.q
    nr = int(g["r_seqs"])
    rlen = int(off[nr])
    n_iter = max(int(k.split("_")[1]) for k in g if k.startswith("v_") and k[2:].isdigit()) + 1
    for it in range(n_iter):
        s = orc.linear_s(v, vbg, c.K, c.W, Kb)
        if f"s_{it}" in g:
            assert np.array_equal(s, g[f"s_{it}"])            # Motif.cpp:485-494
        r, llh = orc.estep(kmer, off, c.K, c.W, s, q)
        assert np.array_equal(r[:rlen], g[f"r_{it}"])         # EM.cpp:149-196
        np.testing.assert_allclose(llh, g[f"llh_{it}"], rtol=1e-6)
        n = orc.mstep_counts(kmer, off, c.K, c.W, r)
        if f"n_{it}" in g:
            assert np.array_equal(n, g[f"n_{it}"])            # EM.cpp:217-254
        v = orc.update_v(n, c.A, vbg, c.K, c.W)
        if f"v_{it}" in g:
            assert np.array_equal(v, g[f"v_{it}"])            # Motif.h:95-136
    assert np.float32(orc.optimize_q(r, off, c.W)) == g["q_after_optimize_q"]      # EM.cpp:505-519
    if "p_final" in g:
        assert np.array_equal(orc.calculate_p(v, vbg, c.bg_order, c.K, c.W), g["p_final"])
        assert np.array_equal(bm.calculate_p(v, vbg, c.bg_order, c.K, c.W), g["p_final"])
    s_log = orc.log_s(v, vbg, c.K, c.W, Kb)
    if "logs_final" in g:
        np.testing.assert_allclose(s_log, g["logs_final"], rtol=0, atol=2e-6)
    mops, zoops, z = orc.logodds(kmer, off, c.K, c.W, s_log)
    np.testing.assert_allclose(zoops, g["zoops"], rtol=0, atol=5e-5)
    assert np.array_equal(z, g["z"])
    if "mops" in g:
        np.testing.assert_allclose(mops, g["mops"], rtol=0, atol=5e-5)
    for oq in (0, 1):
        if f"opt{oq}_v" in g:
            res = orc.optimize(kmer, off, c.K, c.W, c.bg_order, vbg, c.A, c.v0, c.q, optimizeQ=bool(oq))
            assert np.array_equal(res["v"], g[f"opt{oq}_v"])  # EM.cpp:62-137 incl. the stop rule
            assert np.float32(res["q"]) == g[f"opt{oq}_q"]
            np.testing.assert_allclose(res["llh"], g[f"opt{oq}_llh"], rtol=1e-6)
            assert np.array_equal(res["n"], g[f"opt{oq}_n"])


@pytest.mark.parametrize("name", NAMES)
def test_product_host_path_matches_reference_golden(name, lib):
    """FASTA codes -> 2-bit pack (+N exceptions) -> background model, all in the product's own
    host code, against what the reference's Sequence / BackgroundModel produced."""
    c, g = gu.load(name)
    pk = bm.PackedSeqs.from_codes(c.codes, c.in_off, c.ss, seed=42)
    assert np.array_equal(pk.offsets(), g["off"])
    if "kmer" in g:
        for K in (0, 2, 10):
            assert np.array_equal(pk.unpack_y(K).astype(np.uint64), g["kmer"] % np.uint64(4 ** (K + 1)))
    assert np.array_equal(pk.bg_model(c.bg_order, c.alpha_bg), g["vbg"])


@pytest.mark.skipif(not __import__("oracle").have_reference(), reason="reference build absent")
def test_golden_is_fresh_against_the_reference_build():
    """Where oracle/_ref exists, re-run one fixture through the reference and compare."""
    import oracle
    c, g = gu.load("small_k2_ds_N")
    R = oracle.Reference()
    R.set_threads(1)
    S = R.session(c.codes, c.in_off, c.ss, 42)
    assert np.array_equal(S.kmers(), g["kmer"])
    bg, vbg = S.bg(c.bg_order, c.alpha_bg)
    assert np.array_equal(vbg, g["vbg"])
    m = S.motif(c.W, c.K, c.alpha, bg, c.q, c.v0)
    e = S.em(m, bg, False, False)
    S.R.ref_em_estep(e); S.R.ref_em_mstep(e)
    assert np.array_equal(S.motif_v(m), g["v_0"])
    S.close()
