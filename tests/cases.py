"""Seeded test cases shared by the CPU and GPU suites (inputs only; expectations come from the
oracle at run time or from tests/golden)."""
from __future__ import annotations

import numpy as np

from bammmotif2_amd import synth


class Case:
    def __init__(self, name, N, L0, W, K, ss=False, n_frac=0.0, ragged=0, seed=7, bg_order=2, q=0.3,
                 plant_frac=0.5):
        self.name, self.N, self.L0, self.W, self.K = name, N, L0, W, K
        self.ss, self.n_frac, self.ragged, self.seed, self.bg_order, self.q = ss, n_frac, ragged, seed, bg_order, q
        self.pwm = synth.make_pwm(W, seed)
        self.codes, self.in_off = synth.make_sequences(N, L0, self.pwm, seed, plant_frac, n_frac, ragged)
        self.alpha = synth.default_alpha(K)
        self.A = synth.alpha_matrix(self.alpha, W)
        self.alpha_bg = np.array([1.0] + [10.0] * bg_order, np.float32)
        # a slightly blurred seed model so that EM has something to learn
        blur = (0.7 * self.pwm + 0.3 * 0.25).astype(np.float32)
        self.v0 = synth.bamm_from_pwm(blur, K)

    def encode(self, orc):
        """(seq, kmer, off, vbg) through the pinned oracle (Sequence.cpp / BackgroundModel.cpp)."""
        seq, kmer, off = orc.encode_set(self.codes, self.in_off, self.ss, 42)
        vbg = orc.bg_model(kmer, off, self.bg_order, self.alpha_bg)
        return seq, kmer, off, vbg


SMALL_CASES = [
    dict(name="k2_ds_N", N=96, L0=60, W=8, K=2, n_frac=0.02, ragged=10),
    dict(name="k0_ss", N=64, L0=45, W=6, K=0, ss=True, n_frac=0.05, ragged=6),
    dict(name="k3_ds", N=48, L0=70, W=10, K=3, ragged=5),
    dict(name="k1_heavyN", N=40, L0=50, W=7, K=1, n_frac=0.10, ragged=4),
    dict(name="k2_long_ss", N=24, L0=700, W=20, K=2, ss=True, ragged=300),
    dict(name="k2_w1", N=32, L0=40, W=1, K=2, ragged=3),
    dict(name="k2_config2_shape", N=200, L0=200, W=20, K=2),
]


def small_case(i_or_name) -> Case:
    for i, d in enumerate(SMALL_CASES):
        if i == i_or_name or d["name"] == i_or_name:
            return Case(**d)
    raise KeyError(i_or_name)
