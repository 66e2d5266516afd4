"""Seeded synthetic inputs of the shapes BASELINE.json names.

N x L0 i.i.d. uniform ACGT; with probability ``plant_frac`` one site sampled from a W-column
PWM (columns ~ Dirichlet(0.3)) is planted at a uniform start.  numpy's legacy RandomState is
used on purpose: its streams are frozen, so the same seed gives the same bytes everywhere.
Codes follow the reference alphabet encoding (0 = N, 1..4 = A,C,G,T; Alphabet.cpp:36-40).
"""
from __future__ import annotations

import numpy as np


def make_pwm(W: int, seed: int = 1234, sharp: int = 3) -> np.ndarray:
    """[4][W] float32 column-stochastic matrix.

    Columns are integer weights raised to `sharp` (multiplications only): unlike a Dirichlet
    draw this never calls libm, whose exp/log variants differ between CPU generations, so the
    same seed gives bit-identical matrices on every box."""
    rs = np.random.RandomState(seed)
    w = rs.randint(1, 100, size=(W, 4)).astype(np.float64) ** sharp
    cols = w / w.sum(axis=1, keepdims=True)
    cols = np.maximum(cols, 1e-4)
    cols /= cols.sum(axis=1, keepdims=True)
    return np.ascontiguousarray(cols.T.astype(np.float32))


def make_sequences(N: int, L0: int, pwm: np.ndarray, seed: int = 1234, plant_frac: float = 0.5,
                   n_frac: float = 0.0, ragged: int = 0):
    """Returns (codes uint8 [sum L0_n], off uint64 [N+1]).

    ragged > 0 draws each length uniformly from [L0-ragged, L0+ragged]."""
    rs = np.random.RandomState(seed + 1)
    W = pwm.shape[1]
    if ragged:
        lens = rs.randint(L0 - ragged, L0 + ragged + 1, size=N).astype(np.int64)
    else:
        lens = np.full(N, L0, np.int64)
    off = np.zeros(N + 1, np.uint64)
    off[1:] = np.cumsum(lens)
    codes = rs.randint(1, 5, size=int(off[-1])).astype(np.uint8)
    planted = np.nonzero(rs.random_sample(N) < plant_frac)[0]
    if len(planted):
        cdf = np.cumsum(pwm.astype(np.float64), axis=0)      # [4][W]
        u = rs.random_sample((len(planted), W))
        site = (u[:, None, :] > cdf[None, :3, :]).sum(axis=1).astype(np.uint8) + 1   # [P][W] in 1..4
        start = (rs.random_sample(len(planted)) * (lens[planted] - W + 1)).astype(np.int64)
        base = off[planted].astype(np.int64) + start
        idx = base[:, None] + np.arange(W)[None, :]
        codes[idx.ravel()] = site.ravel()
    if n_frac > 0:
        codes[rs.random_sample(len(codes)) < n_frac] = 0
    return codes, off


def bamm_from_pwm(pwm: np.ndarray, K: int) -> np.ndarray:
    """Deterministic order-K seed: every order repeats the PWM column (no context dependence).

    Flat [k][y][j] layout of include/bamm_em.h."""
    W = pwm.shape[1]
    parts = []
    for k in range(K + 1):
        reps = 4 ** k
        parts.append(np.tile(pwm, (reps, 1)) if False else np.repeat(pwm[None, :, :], reps, axis=0).reshape(reps * 4, W))
    # row index y = ctx*4 + base  (newest base least significant, Sequence.cpp:35-41)
    return np.ascontiguousarray(np.concatenate([p.ravel() for p in parts]).astype(np.float32))


def default_alpha(K: int, beta: float = 7.0, gamma: float = 3.0) -> np.ndarray:
    """alpha_k = {1, beta*gamma^k} (Global.cpp:35-38,225-232)."""
    return np.array([1.0] + [beta * gamma ** k for k in range(1, K + 1)], np.float32)


def alpha_matrix(alpha: np.ndarray, W: int) -> np.ndarray:
    """A[k][j] = alpha_k (Motif.cpp:43-46)."""
    return np.ascontiguousarray(np.repeat(np.asarray(alpha, np.float32), W))
