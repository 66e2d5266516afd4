"""The oracle's restatement of EM::mask (EM.cpp:261-503) against what the real reference produced
(tests/golden/mask_*.npz, written by tests/golden/make_golden.py from oracle/_ref).  Everything is
serial fp32 in the reference, so the restatement has to match bit for bit."""
import glob
import hashlib
import os

import numpy as np
import pytest

from tests.cases import SMALL_CASES, Case
from tests.golden_util import GOLDEN_DIR

NAMES = sorted(os.path.basename(p)[5:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "mask_*.npz")))


def test_mask_fixtures_present():
    assert len(NAMES) >= 6, NAMES


@pytest.mark.parametrize("name", NAMES)
def test_oracle_mask_matches_reference_golden(name, orc):
    c = Case(**next(d for d in SMALL_CASES if d["name"] == name))
    g = np.load(os.path.join(GOLDEN_DIR, f"mask_{name}.npz"))
    seq, kmer, off, vbg = c.encode(orc)
    assert hashlib.sha256(np.ascontiguousarray(kmer).tobytes()).hexdigest() == str(g["kmer_sha256"])
    assert np.array_equal(vbg, g["vbg"])
    for oq in (0, 1):
        for f in (0.05, 0.2):
            t = f"oq{oq}_f{int(f * 100)}"
            res = orc.mask(kmer, off, c.K, c.W, c.bg_order, vbg, c.A, c.v0, c.q, optimizeQ=bool(oq), f=f)
            assert np.array_equal(res["v"], g[t + "_v"])
            assert np.array_equal(res["n"], g[t + "_n"])
            assert np.float32(res["q"]) == g[t + "_q"]
            np.testing.assert_allclose(np.float32(res["llh"]), g[t + "_llh"], rtol=1e-6)    # logf variants
            assert hashlib.sha256(res["r"].tobytes()).hexdigest() == str(g[t + "_r_sha256"])
            assert res["iterations"] == int(g[t + "_iterations"])
            assert np.float32(res["cutoff"]) == g[t + "_cutoff"] and res["listed"] == int(g[t + "_listed"])
            assert np.array_equal(orc.calculate_p(res["v"], vbg, c.bg_order, c.K, c.W), g[t + "_p"])


def test_live_reference_mask_if_built(orc):
    """Where oracle/_ref exists (the development container) run the reference's mask() itself."""
    import oracle
    if not oracle.have_reference():
        pytest.skip("reference build not present")
    R = oracle.Reference()
    R.set_threads(1)
    c = Case(**SMALL_CASES[3])
    S = R.session(c.codes, c.in_off, c.ss, 42)
    bg, vbg = S.bg(c.bg_order, c.alpha_bg)
    m = S.motif(c.W, c.K, c.alpha, bg, c.q, c.v0)
    e = S.em(m, bg, True, False, 0.1)
    S.R.ref_em_mask(e)
    res = orc.mask(S.kmers(), S.off, c.K, c.W, c.bg_order, vbg, c.A, c.v0, c.q, optimizeQ=True, f=0.1)
    assert np.array_equal(res["v"], S.motif_v(m)) and np.array_equal(res["r"], S.em_r(e))
    assert np.float32(res["q"]) == np.float32(S.R.ref_em_q(e))
    S.close()
