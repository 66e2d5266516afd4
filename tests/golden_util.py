"""Load tests/golden fixtures (expectations produced by the real reference, see
tests/golden/make_golden.py) and rebuild the matching inputs."""
from __future__ import annotations

import glob
import hashlib
import os

import numpy as np

from tests.cases import SMALL_CASES, Case

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
LARGE_SPECS = {
    "g4_1k": dict(name="g4_1k", N=1000, L0=200, W=20, K=2, seed=1234),
    "g4_10k": dict(name="g4_10k", N=10000, L0=200, W=20, K=2, seed=1234),
    "g5_k4": dict(name="g5_k4", N=300, L0=500, W=30, K=4, seed=77, ss=True),
    "c4_k4_ds": dict(name="c4_k4_ds", N=160, L0=500, W=30, K=4, seed=78),
}


def digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def fixture_names():
    """EM-path fixtures (small_* / large_*); eval_small / travis_jund have their own tests."""
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz"))
                  if os.path.basename(p).startswith(("small_", "large_")))


def load(name):
    g = dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz")))
    short = name.split("_", 1)[1]
    if name.startswith("small_"):
        spec = next(d for d in SMALL_CASES if d["name"] == short)
    else:
        spec = LARGE_SPECS[short]
    c = Case(**spec)
    if "codes" in g:                       # stored inputs win over regenerated ones
        c.codes, c.in_off = g["codes"], g["in_off"]
    assert digest(c.codes) == str(g["codes_sha256"]), "synthetic input generator drifted"
    assert digest(c.in_off) == str(g["in_off_sha256"])
    for k in ("v0", "A", "alpha", "alpha_bg"):
        assert np.array_equal(getattr(c, k), g[k]), k
    return c, g
