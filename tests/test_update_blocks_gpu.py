"""The model update spread over blocks (k_update_counts + k_update_model, csrc/model.hip) against the one-block
k_update<false>: tables beyond the update's LDS form (k >= 3 at usual widths, k = 2 at W > 32, orders 7-10).

Both restate EM.cpp:247-254 (lower-order counts: four rows of the next order, ascending) and Motif.h:95-136 (the
interpolated conditionals) with the same float expressions on the same integers, so counts, model, odds table, q and
the log-likelihood are IDENTICAL bits; v_diff is an fp64 sum of |dv| whose grouping follows the launch shape.
"""
import numpy as np
import pytest

import bammmotif2_amd as bm
from bammmotif2_amd import synth

pytestmark = pytest.mark.gpu

SHAPES = [
    dict(N=3000, L0=200, W=20, K=3),
    dict(N=1500, L0=300, W=30, K=4, n_frac=0.01, ragged=40),
    dict(N=1200, L0=150, W=12, K=5, ss=True),
    dict(N=2000, L0=120, W=40, K=2),                       # 16 x 40 cells: too many for the LDS form
    dict(N=600, L0=200, W=8, K=7),                         # tables in global memory (orders 7-10)
    dict(N=900, L0=100, W=6, K=6),
]


def run(ctx, blocks, N, L0, W, K, ss=False, n_frac=0.0, ragged=0, seed=9):
    pwm = synth.make_pwm(W, seed)
    codes, in_off = synth.make_sequences(N, L0, pwm, seed, 0.5, n_frac, ragged)
    packed = bm.PackedSeqs.from_codes(codes, in_off, ss, seed=42)
    A = synth.alpha_matrix(synth.default_alpha(K), W)
    vbg = packed.bg_model(2, np.array([1.0, 10.0, 10.0], np.float32))
    v0 = synth.bamm_from_pwm((0.7 * pwm + 0.075).astype(np.float32), K)
    ctx.set_tuning(update_blocks=int(blocks))
    try:
        seqs = bm.SeqSet(ctx, packed)
        em = bm.EM(ctx, seqs, K, W, vbg, A, v0, 0.3, optimizeQ=True, max_iterations=60, epsilon=0.01)
    finally:
        ctx.set_tuning(update_blocks=1)
    em.iterate(3)
    em.EStep(); em.MStep(); em.optimize_q()                 # the stepwise calls share the update
    first = (em.getV(), em.getCounts(), em.getS(), em.getQ(), em.getLLH(), em.iteration())
    it = em.optimize()
    out = first + (it, em.getV(), em.getCounts(), em.getS(), em.getQ(), em.trace(), em.getR(0, 40))
    em.close(); seqs.close()
    return out


@pytest.mark.parametrize("shape", SHAPES, ids=[f"K{d['K']}_W{d['W']}_N{d['N']}" for d in SHAPES])
def test_update_over_blocks_equals_one_block(shape, gpu_ctx):
    a = run(gpu_ctx, True, **shape)
    b = run(gpu_ctx, False, **shape)
    for i in (0, 1, 2, 7, 8, 9, 12):
        assert np.array_equal(a[i], b[i]), i
    assert a[3] == b[3] and a[4] == b[4] and a[5] == b[5] and a[6] == b[6] and a[10] == b[10]
    (la, va, qa), (lb, vb, qb) = a[11], b[11]
    assert np.array_equal(la, lb) and np.array_equal(qa, qb)
    np.testing.assert_allclose(va, vb, rtol=3e-7, atol=0)
    assert 1 < a[6] <= 60


def test_scratch_blocks_pass_from_handle_to_handle(gpu_ctx):
    """The context keeps a closed handle's set-sized scratch for the next one (csrc/abi.cpp: scratch_alloc): same
    results from a block that is fresh, reused, reused after a LARGER owner, and with the cache turned off."""
    shape = dict(N=6000, L0=400, W=24, K=4, n_frac=0.01, ragged=60)      # sliced path: dense r, lists
    small = dict(N=3000, L0=300, W=24, K=4)
    ref = run(gpu_ctx, True, **shape)
    again = run(gpu_ctx, True, **shape)                      # takes over the first run's blocks (poisoned by the fixture)
    run(gpu_ctx, True, **small)
    third = run(gpu_ctx, True, **shape)
    gpu_ctx.set_tuning(scratch_cache_mb=0)
    try:
        plain = run(gpu_ctx, True, **shape)
    finally:
        gpu_ctx.set_tuning(scratch_cache_mb=16384)
    for other in (again, third, plain):
        for i in (0, 1, 2, 7, 8, 9, 12):
            assert np.array_equal(ref[i], other[i]), i
        assert ref[6] == other[6] and ref[10] == other[10]
