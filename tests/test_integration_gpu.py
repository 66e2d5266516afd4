"""The drop-in boundary, exercised: the reference's OWN classes (`EM`, `ScoreSeqSet`, `FDR`, `Motif`,
`BackgroundModel`, `Sequence`, compiled from /root/reference/src by `make -C oracle ref_hip`) with
integration/EM_hip.cpp in place of src/refinement/EM.cpp and integration/ScoreSeqSet_hip.cpp's calcLogOdds in
place of the reference's, i.e. with their hot path forwarded to libbamm_em.so -- driven through the same
harness (oracle/ref_harness.cpp) that produced tests/golden from the unmodified reference.  EM.h and
ScoreSeqSet.h are the reference's, unchanged.

The shared object is built in the development container (the reference tree does not travel) and rides to
the GPU box like oracle/_ref/libbammref.so."""
import os

import numpy as np
import pytest

import oracle
from tests import golden_util as gu
from tests import margins

pytestmark = pytest.mark.gpu

HIP_REF = os.path.join(os.path.dirname(oracle.REF_SO), "libbammref_hip.so")


@pytest.fixture(scope="module")
def R(lib):
    if not os.path.exists(HIP_REF):
        pytest.skip("oracle/_ref/libbammref_hip.so not built (make -C oracle ref_hip, development container only)")
    r = oracle.Reference(HIP_REF)
    r.set_threads(1)
    return r


@pytest.mark.parametrize("name", ["small_k2_ds_N", "small_k0_ss", "small_k3_ds", "small_k1_heavyN"])
def test_reference_em_class_on_the_gpu_reproduces_the_goldens(name, R):
    """EM::EStep / MStep / optimize_q / getR / optimize of the reference's class EM (EM.h:20-36), running on the
    MI355X, against the vectors the unmodified reference produced."""
    c, g = gu.load(name)
    S = R.session(c.codes, c.in_off, c.ss, 42)
    assert gu.digest(S.kmers()) == str(g["kmer_sha256"])          # Sequence.cpp is the reference's own
    bg, vbg = S.bg(c.bg_order, c.alpha_bg)
    assert np.array_equal(vbg, g["vbg"])
    m = S.motif(c.W, c.K, c.alpha, bg, c.q, c.v0)
    e = S.em(m, bg, False, False)
    nr = int(g["r_seqs"])
    rlen = int(S.off[nr])
    for it in range(3):
        S.R.ref_em_estep(e)
        if it == 0:
            assert np.array_equal(S.motif_s(m, c.K, c.W), g["s_0"])                    # Motif.cpp:485-494, bit-exact
        fl = "reference EM class over the C ABI"
        margins.check(name, fl, f"r pass {it + 1}", S.em_r(e)[:rlen], g[f"r_{it}"], 1e-5, 1e-12)
        np.testing.assert_allclose(np.float32(S.R.ref_em_llh(e)), g[f"llh_{it}"], rtol=1e-5)
        S.R.ref_em_mstep(e)
        margins.check(name, fl, f"n pass {it + 1}", S.em_n(e, c.K, c.W), g[f"n_{it}"], 1e-5, 1e-6)
        margins.check(name, fl, f"v pass {it + 1}", S.motif_v(m), g[f"v_{it}"], 1e-5, 1e-9)   # flat: observed <= 2.2e-6 over three passes
    S.R.ref_em_optimize_q(e)
    np.testing.assert_allclose(np.float32(S.R.ref_em_q(e)), g["q_after_optimize_q"], rtol=1e-5)
    # the full loop, stopping rule included (EM.cpp:62-137); the model lands in the caller's Motif
    for oq in (0, 1):
        m2 = S.motif(c.W, c.K, c.alpha, bg, c.q, c.v0)
        e2 = S.em(m2, bg, bool(oq), False)
        S.R.ref_em_optimize(e2)
        np.testing.assert_allclose(np.float32(S.R.ref_em_llh(e2)), g[f"opt{oq}_llh"], rtol=2e-4)
        np.testing.assert_allclose(S.motif_v(m2), g[f"opt{oq}_v"], rtol=5e-3, atol=1e-6)     # +-1 iteration at the knife edge
        np.testing.assert_allclose(np.float32(S.R.ref_em_q(e2)), g[f"opt{oq}_q"], rtol=1e-3)
    S.close()


def test_reference_scoreseqset_on_the_gpu_is_bit_exact(R):
    """ScoreSeqSet::calcLogOdds through the library: MOPS / ZOOPS scores and arg-max positions of the golden's
    final model, bit for bit (the log-odds table is the reference's own calculateLogS)."""
    c, g = gu.load("small_k2_ds_N")
    S = R.session(c.codes, c.in_off, c.ss, 42)
    bg, _ = S.bg(c.bg_order, c.alpha_bg)
    m = S.motif(c.W, c.K, c.alpha, bg, c.q, g["v_2"])
    mops, zoops, z = S.logodds(m, bg, c.W)
    assert np.array_equal(zoops, g["zoops"]) and np.array_equal(z, g["z"]) and np.array_equal(mops, g["mops"])
    S.close()


def test_reference_fdr_class_on_the_gpu(R):
    """FDR::evaluateMotif (FDR.cpp:28-145) constructs one EM and two ScoreSeqSet objects per fold: with the
    swapped implementation files every one of them runs on the GPU; statistics and files as the unmodified
    reference wrote them (tests/golden/eval_small.npz), up to near-tie swaps along the ranking."""
    g = dict(np.load(os.path.join(gu.GOLDEN_DIR, "eval_small.npz")))
    from tests.cases import Case
    c = Case("eval", N=120, L0=50, W=8, K=1, seed=11, n_frac=0.01)
    S = R.session(c.codes, c.in_off, c.ss, 42)
    bg, vbg = S.bg(c.bg_order, c.alpha_bg)
    neg = S.negset(2, 2, False)
    assert np.array_equal(neg.seq_codes(), g["neg_codes"])
    m = S.motif(c.W, c.K, c.alpha, bg, c.q, c.v0)
    files, scores, q = S.fdr(neg, m, bg, 4, True, True, em=True, optimizeQ=False, threads=2)
    for mine, ref in zip(scores[:2], (g["fdr_pos_max"], g["fdr_neg_max"])):
        np.testing.assert_allclose(np.sort(mine), np.sort(ref), rtol=1e-3, atol=2e-3)
    ref_rows = g["fdr_file_zoops_stats"].tobytes().decode().strip().split("\n")
    my_rows = files["zoops.stats"].decode().strip().split("\n")
    assert len(my_rows) == len(ref_rows) and my_rows[0].split("\t")[:6] == ref_rows[0].split("\t")[:6]
    a = np.array([r.rstrip("\t").split("\t") for r in my_rows[1:]], float)
    b = np.array([r.rstrip("\t").split("\t") for r in ref_rows[1:]], float)
    assert np.mean(np.all(a[:, :2] == b[:, :2], axis=1)) > 0.97
    S.close()
